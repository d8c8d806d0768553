// extern "C" door into the reference's vendored hilbert_c2i (cpp/src/vendored/hilbert.cpp:196-237) so that
// tests can call it through ctypes.  Compiled together with the reference file where it lies; this shim
// contains no reference code.  TEST INFRASTRUCTURE ONLY.
#include "hilbert.h"
extern "C" unsigned long long ref_hilbert_c2i_3d(unsigned nbits, unsigned long long a, unsigned long long b, unsigned long long c) {
    bitmask_t coord[3] = {a, b, c};
    return (unsigned long long)hilbert_c2i(3, nbits, coord);
}
extern "C" void ref_hilbert_lut(unsigned grid, unsigned nbits, unsigned int *out) {
    for (unsigned i = 0; i < grid; i++)
        for (unsigned j = 0; j < grid; j++)
            for (unsigned k = 0; k < grid; k++) {
                bitmask_t coord[3] = {i, j, k};
                out[(i * grid + j) * grid + k] = (unsigned int)hilbert_c2i(3, nbits, coord);
            }
}
