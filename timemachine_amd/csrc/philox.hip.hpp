// Philox4x32-10 counter-based generator (Salmon et al., SC'11): the noise source of the integrator and the barostat.
// Stateless: (counter, key) -> 128 random bits, so kernels draw their own numbers and runs are reproducible per seed.
#pragma once
#include "common.hpp"

namespace tmamd {

__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0, unsigned int k1, unsigned int out[4]) {
    const unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = static_cast<unsigned long long>(M0) * c0;
        const unsigned long long p1 = static_cast<unsigned long long>(M1) * c2;
        const unsigned int n0 = static_cast<unsigned int>(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = static_cast<unsigned int>(p1);
        const unsigned int n2 = static_cast<unsigned int>(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = static_cast<unsigned int>(p0);
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

} // namespace tmamd
