// Cost of u64 global atomic adds by memory scope on MI355X: agent scope (executed at the memory side: the XCD L2s are not
// coherent with each other) vs workgroup scope (executed in the issuing XCD's L2).  The pattern imitates the tile kernel's
// flush: 3072 waves, each issuing rounds of 64-lane atomics to 3 consecutive u64 per "atom", atoms spread over a
// 23.5k-atom accumulator (one accumulator per XCD in the local variant).
//   hipcc --offload-arch=gfx950 -O3 atomic_scope.hip -o atomic_scope && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPE, bool PER_XCD, bool SOA = false, bool SCATTER = false>
__global__ void k_flush(unsigned long long *acc, const int n_atoms, const int rounds, unsigned int *xcc_seen) {
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; // XCC_ID
    if (lane == 0) {
        xcc_seen[wave] = xcc;
    }
    unsigned long long *base = PER_XCD ? acc + static_cast<size_t>(xcc) * n_atoms * 3 : acc;
    unsigned int h = wave * 2654435761u + 12345u;
    for (int r = 0; r < rounds; r++) {
        h = h * 1664525u + 1013904223u;
        unsigned int atom = ((h >> 8) % static_cast<unsigned int>(n_atoms - 64)) + lane; // 64 consecutive atoms
        if (SCATTER) {
            atom = (atom * 2654435761u) % static_cast<unsigned int>(n_atoms); // no locality at all
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            unsigned long long *p = SOA ? base + static_cast<size_t>(c) * n_atoms + atom : base + static_cast<size_t>(atom) * 3 + c;
            __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, SCOPE);
        }
    }
}

// lanes hit u64 words `stride_words` apart (8 = one word per 64-byte line), same count of atomics
template <int STRIDE_WORDS>
__global__ void k_stride(unsigned long long *acc, const int n_atoms, const int rounds, unsigned int *) {
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned int h = wave * 2654435761u + 12345u;
    const unsigned int span = 64u * STRIDE_WORDS;
    for (int r = 0; r < rounds * 3; r++) {
        h = h * 1664525u + 1013904223u;
        const unsigned int start = ((h >> 8) % (static_cast<unsigned int>(n_atoms) * 24u - span)) & ~15u; // 128-byte aligned
        __hip_atomic_fetch_add(acc + start + lane * STRIDE_WORDS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int n_atoms = 23559, waves = 3072, rounds = 6; // 3072 * 6 * 64 * 3 = 3.5 M atomics
    unsigned long long *acc;
    unsigned int *xcc;
    CK(hipMalloc(&acc, sizeof(unsigned long long) * n_atoms * 3 * 8));
    CK(hipMalloc(&xcc, sizeof(unsigned int) * waves));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kernel) -> int {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            CK(hipMemset(acc, 0, sizeof(unsigned long long) * n_atoms * 3 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            kernel<<<waves / 12, 768>>>(acc, n_atoms, rounds, xcc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        std::vector<unsigned long long> h(static_cast<size_t>(n_atoms) * 3 * 8);
        CK(hipMemcpy(h.data(), acc, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long total = 0;
        for (auto v : h) total += v;
        printf("%-34s %8.1f us   sum %llu (expected %llu)\n", name, best * 1e3f, total, static_cast<unsigned long long>(waves) * rounds * 64 * 3);
        return 0;
    };
    if (run("agent scope, one accumulator", k_flush<__HIP_MEMORY_SCOPE_AGENT, false>)) return 1;
    if (run("agent scope, per-XCD accumulators", k_flush<__HIP_MEMORY_SCOPE_AGENT, true>)) return 1;
    if (run("workgroup scope, per-XCD", k_flush<__HIP_MEMORY_SCOPE_WORKGROUP, true>)) return 1;
    if (run("wavefront scope, per-XCD", k_flush<__HIP_MEMORY_SCOPE_WAVEFRONT, true>)) return 1;
    if (run("agent, component-major (SoA)", k_flush<__HIP_MEMORY_SCOPE_AGENT, false, true>)) return 1;
    if (run("agent, scattered atoms (AoS)", k_flush<__HIP_MEMORY_SCOPE_AGENT, false, false, true>)) return 1;
    if (run("agent, scattered atoms (SoA)", k_flush<__HIP_MEMORY_SCOPE_AGENT, false, true, true>)) return 1;
    if (run("dense: 64 lanes x 8 B (512 B)", k_stride<1>)) return 1;
    if (run("stride 16 B (1 KB)", k_stride<2>)) return 1;
    if (run("stride 32 B (2 KB)", k_stride<4>)) return 1;
    if (run("stride 64 B (4 KB)", k_stride<8>)) return 1;
    if (run("stride 128 B (8 KB)", k_stride<16>)) return 1;
    if (run("stride 256 B (16 KB)", k_stride<32>)) return 1;
    std::vector<unsigned int> hx(waves);
    CK(hipMemcpy(hx.data(), xcc, waves * 4, hipMemcpyDeviceToHost));
    int cnt[16] = {0};
    for (auto v : hx) cnt[v & 15]++;
    printf("waves per XCC:");
    for (int k = 0; k < 8; k++) printf(" %d", cnt[k]);
    printf("\n");
    return 0;
}
