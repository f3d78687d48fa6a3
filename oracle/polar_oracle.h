/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's float polar decoders, written from the
 * behavioural spec in SURVEY.md App. A and the reference sources cited per function in
 * polar_oracle.c.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product (polar-codes_amd/) never does.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this oracle against golden vectors produced
 * by the unmodified reference library (oracle/_ref, see oracle/gen_golden.sh) and against the
 * reference's own known-answer tests (decodingtest.cpp, qa_pypolar_detector.py).
 */
#ifndef POLAR_ORACLE_H
#define POLAR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error detector ids (reference: errordetection/errordetector.cpp:27-88) */
enum {
    PO_DET_DUMMY = 0,
    PO_DET_CRC8 = 8,
    PO_DET_CRC11NR = 11,
    PO_DET_CRC24NRC = 24,
    PO_DET_CRC32 = 32
};

/* returns 1/0; mirrors Detector::check(void*, int bytes) */
int po_det_check(int det, const uint8_t* data, int bytes);
/* mirrors Detector::generate(void*, int bytes): writes the checksum into the tail */
void po_det_generate(int det, uint8_t* data, int bytes);
/* raw checksum over `bytes` message bytes (CRC++ CalculateBits / crc32.cpp gen / crc8.cpp gen) */
uint32_t po_det_calc(int det, const uint8_t* data, int bytes);

/* F_function / G_function / CombineBits* on plain arrays (avx_float.h:101-236), for KATs */
void po_f(const float* in, float* out, unsigned h);
void po_g(const float* in, const float* bits, float* out, unsigned h);

/* Fast-SSC float decoder (fastssc_avx_float.cpp).  bits_out: N sign bits (0/1).
 * returns 0, or -1 if the reference would throw std::invalid_argument. */
int po_sc_decode_bits(unsigned N, const uint32_t* frozen, unsigned nF, const float* llr,
                      uint8_t* bits_out);

/* Fast-SSCL float decoder (scl_avx_float.cpp).  bits_out: npaths x N sign bits in final list
 * order, metrics: npaths floats (best first).  returns the final path count. */
int po_scl_decode_paths(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF,
                        const float* llr, uint8_t* bits_out, float* metrics);

/* debugging aid: after every leaf of the SCL tree, (1 + L) floats {path count, metrics...};
 * returns the number of leaves recorded (<= cap). */
int po_scl_trace(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF, const float* llr,
                 float* trace, int cap);

/* Decoder::decode_vector for type "float" (L==1 -> Fast-SSC, else Fast-SSCL) and "mixed"
 * (adaptive_float.cpp:33-45).  dectype: 0 float, 2 mixed.  out: (K+7)/8 bytes per row,
 * ok: 1 byte per row, metrics (nullable): L floats per row (best first, 0 padded),
 * returns 0 or a negative error. */
int po_decode_batch(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF, int det,
                    int systematic, int dectype, const float* llr, size_t rows, uint8_t* out,
                    uint8_t* ok, float* metrics);

/* FastSscFipChar (fastssc_fip_char.cpp): hard decisions of the N code bits for one row of int8 LLRs.
 * (po_decode_batch with dectype 3 runs the int8 decoders: FastSscFipChar for L <= 1, SclFipChar above; the rows are
 * passed as int8-valued floats.) */
int po_scc_decode_bits(unsigned N, const uint32_t* frozen, unsigned nF, const int8_t* llr, uint8_t* bits_out);

/* ButterflyFipPacked::encode_vector (butterfly_fip_packed.cpp:45-59): info (K+7)/8 bytes in
 * (checksum written into its tail, as the reference does), codeword N/8 bytes out, MSB first. */
void po_encode(unsigned N, const uint32_t* frozen, unsigned nF, int det, int systematic,
               uint8_t* info, uint8_t* codeword);

/* node statistics of the decode tree the reference would build (createDecoder of each decoder):
 * counts[0]=leaves, counts[1]=inner nodes.  list!=0 -> SclAvx::createDecoder, else FastSscAvx. */
int po_tree_stats(unsigned N, const uint32_t* frozen, unsigned nF, int list, unsigned counts[2]);

#ifdef __cplusplus
}
#endif
/* ---- Puncturer (puncturer.cpp:51-89, puncturer.h:66-97) ------------------------------------------------------
 * positions[M]: the parent-block positions copied to the output = {0..N-1} minus the first N - M frozen positions
 * (N = M rounded up to a power of two).  Returns 0, or -1 if N - M exceeds nF (the reference throws). */
int po_punc_positions(unsigned M, const uint32_t* frozen, unsigned nF, uint32_t* positions, unsigned* parent);
void po_puncture_f32(const uint32_t* positions, unsigned M, const float* in, float* out);
void po_depuncture_f32(const uint32_t* positions, unsigned M, unsigned N, const float* in, float* out);
void po_puncture_packed(const uint32_t* positions, unsigned M, const uint8_t* in, uint8_t* out);

#endif
