/*
 * polar_amd_diag.h -- diagnostics of the DIAGNOSTIC build of the library (`make -C polar-codes_amd diag` ->
 * lib/libpolar_amd_diag.so): per-leaf traces and per-instruction cycle stamps of the decode kernels.  The release library
 * (lib/libpolar_amd.so, include/polar_amd.h) neither declares nor exports these entry points.
 */
#ifndef POLAR_AMD_DIAG_H
#define POLAR_AMD_DIAG_H

#include "polar_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics: decode ONE row (host float32[N]) with the list decoder and record, after every leaf of
 * the decode tree, (1 + list_size) floats {live path count, path metrics...} into trace (host). */
int pcd_scl_trace(pcd_decoder* d, const float* llr_row, float* trace, unsigned cap);

/* Diagnostics: shader-clock cycles one wave spends in every instruction of the SCL decode program while
 * decoding ONE row on an otherwise idle GPU (ops[i]/stages[i] describe instruction i; *count = program length). */
int pcd_scl_profile(pcd_decoder* d, const float* llr_row, uint8_t* ops, uint8_t* stages, uint32_t* cycles,
                    unsigned cap, unsigned* count);

/* the same for the SC decode program (16 lanes per codeword, i.e. N <= 2048) */
int pcd_sc_profile(pcd_decoder* d, const float* llr_row, uint8_t* ops, uint8_t* stages, uint32_t* cycles,
                   unsigned cap, unsigned* count);


#ifdef __cplusplus
}
#endif
#endif /* POLAR_AMD_DIAG_H */
