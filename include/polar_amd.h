/*
 * polar_amd.h -- C ABI of the MI355X-native batched polar SC / SCL decoder (pcd_*), encoder (pce_*) and puncturer (ppu_*).
 *
 * Drop-in boundary for the reference's decode path (ant-uni-bremen/polar-codes):
 *   PolarCode::Decoding::create()            src/polarcode/decoding/decoder.cpp:26-52
 *   Decoder::setErrorDetection / setSystematic  decoder.cpp:101-111, include/.../decoder.h:120-143
 *   Decoder::decode_vector(const float*, void*) decoder.cpp:154-167
 *   Decoder::decode_vector(const char*,  void*) decoder.cpp:169-181
 *   pypolar.PolarDecoder                      python/bindings/decoder_python.cc:17-75
 * The reference decodes ONE codeword per call on the CPU; this library decodes B codewords per
 * call on one GPU (one HIP kernel launch), bit-exact with the reference's float decoders
 * (FastSscAvxFloat / SclAvxFloat / AdaptiveFloat) started from a fresh decoder state per codeword.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All functions return 0 (PCD_OK) or a
 * negative pcd_status; pcd_last_error() gives the message of the calling thread's last failure.
 * There is NO CPU fallback: decoding without a usable HIP device fails with PCD_ERR_NO_DEVICE.
 */
#ifndef POLAR_AMD_H
#define POLAR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcd_decoder pcd_decoder; /* opaque handle; not thread-safe (like the reference's Decoder) */

typedef enum {
    PCD_OK = 0,
    PCD_ERR_INVALID_ARG = -1,   /* bad N / L / frozen set / null pointer                              */
    PCD_ERR_UNKNOWN_TYPE = -2,  /* reference: std::logic_error("Unknown PolarDecoder type!")          */
    PCD_ERR_UNSUPPORTED = -3,   /* "scan" decoder, CMAC / CRC-6 / CRC-16: not on this path              */
    PCD_ERR_FROZEN_SET = -4,    /* reference: std::invalid_argument from FastSscAvx::createDecoder     */
    PCD_ERR_CRC_SIZE = -5,      /* reference: std::logic_error("CRC INVALID SIZE!")                    */
    PCD_ERR_UNKNOWN_DETECTOR = -6, /* reference: std::runtime_error("Unknown Error detector requested!") */
    PCD_ERR_NO_DEVICE = -7,     /* no HIP device / HIP runtime error: the product path never falls back */
    PCD_ERR_HIP = -8,
    PCD_ERR_CAPACITY = -9,      /* (N, L) exceeds what the kernels can hold            */
    PCD_ERR_PUNCTURE = -10      /* reference: std::out_of_range from Puncturer::Puncturer               */
} pcd_status;

/* memory-space tags for the pointers handed to pcd_decode_* */
enum { PCD_MEM_HOST = 0, PCD_MEM_DEVICE = 1 };

/* Decoding::create(blockLength, listSize, frozenBits, decoderType)  decoder.cpp:26-52.
 * type is matched like the reference (lower-cased substring): "char" (the int8 decoders FastSscFipChar /
 * SclFipChar, the factory's default branch decoder.cpp:65-67,80-82), "float", "mixed"; "scan" is recognised
 * but PCD_ERR_UNSUPPORTED.  listSize < 2 with a non-char type selects float SC (decoder.cpp:48-50).  frozen[] must be strictly ascending (what frozen_bits() returns).
 * The new decoder carries the CRC-8 detector, as makeDecoder does (decoder.cpp:85).
 * device: HIP device ordinal used by later decode calls (nothing touches the GPU here). */
int pcd_create(size_t block_length, size_t list_size, const uint32_t* frozen, size_t n_frozen,
               const char* type, int device, pcd_decoder** out);
void pcd_destroy(pcd_decoder* d);

/* ErrorDetection::create(size, type)  errordetector.cpp:27-88 + Decoder::setErrorDetection.
 * Supported: type containing "crc" with size 0 (dummy), 8, 11, 24, 32.  */
int pcd_set_error_detection(pcd_decoder* d, unsigned size, const char* type);
/* Decoder::setSystematic / isSystematic (default true, decoder.cpp:92) */
int pcd_set_systematic(pcd_decoder* d, int systematic);
int pcd_is_systematic(const pcd_decoder* d);

size_t pcd_block_length(const pcd_decoder* d);                 /* Decoder::blockLength()  */
size_t pcd_info_length(const pcd_decoder* d);                  /* Decoder::infoLength()   */
size_t pcd_list_size(const pcd_decoder* d);                    /* Decoder::getListSize()  */
size_t pcd_frozen_bits(const pcd_decoder* d, uint32_t* out, size_t cap); /* Decoder::frozenBits() */
/* Decoder::getErrorDetectionMode(): "<TYPE>-<bits>", e.g. "CRC-8", "CRCNR-24", "DUMMY-0" */
int pcd_error_detection_mode(const pcd_decoder* d, char* buf, size_t cap);
size_t pcd_out_bytes(const pcd_decoder* d);                    /* (K + 7) / 8             */

/* Batched Decoder::decode_vector(const float*, void*).
 *   llr     : rows x N float32, row-major               (mem = PCD_MEM_HOST | PCD_MEM_DEVICE)
 *   out     : rows x pcd_out_bytes() decoded bytes       (same memory space as llr)
 *   ok      : rows x uint8, the detector verdict = decode()'s return value          (nullable)
 *   metrics : rows x list_size float32 final path metrics, best first (SCL only)    (nullable)
 *   stream  : hipStream_t (NULL = default stream).  Device pointers: asynchronous on `stream`;
 *             host pointers: staged through device buffers and synchronised before returning.
 *             (The "mixed" decoder is asynchronous as well: the list pass takes the number of codewords
 *             the SC pass failed on from device memory.) */
int pcd_decode_f32(pcd_decoder* d, const float* llr, size_t rows, uint8_t* out, uint8_t* ok,
                   float* metrics, int mem, void* stream);
/* Batched Decoder::decode_vector(const char*, void*): float decoders widen the int8 LLRs exactly as
 * FloatContainer::insertLlr(const char*) does (bitcontainer.cpp:202-207); the int8 ("char") decoders take them as
 * they are (CharContainer::insertLlr(const char*) bitcontainer.cpp:518).  pcd_decode_f32 on an int8 decoder converts
 * like CharContainer::insertLlr(const float*) :505-516 (round to nearest even, saturate to [-128, 127]).  Path metrics
 * of the int8 list decoder are exact integers (the reference's `long`, scl_fip_char.cpp:155) returned as float. */
int pcd_decode_i8(pcd_decoder* d, const int8_t* llr, size_t rows, uint8_t* out, uint8_t* ok,
                  float* metrics, int mem, void* stream);

/* Small batches.  The reference's interface is ONE codeword per call (Decoder::decode_vector decoder.cpp:154-181,
 * decoder_python.cc:41-74); the throughput kernels pack 32 (SC) / 8 .. 2 (list) codewords into a wavefront, whose run time does not
 * depend on how many of its slots are filled.  Calls of up to pcd_small_batch_rows() rows therefore run on a sibling geometry with
 * one codeword per wavefront (shorter per-call latency, same results bit for bit); 0 = no such sibling for this decoder.
 * pcd_set_small_batch_rows changes the threshold (0: every call takes the throughput kernels).  While pcd_set_timing is on, every
 * call takes the throughput kernels as well. */
int pcd_set_small_batch_rows(pcd_decoder* d, size_t rows);
size_t pcd_small_batch_rows(const pcd_decoder* d);

/* Introspection of the compiled decode program (host only, no GPU): counts[0] = leaf ops,
 * counts[1] = inner nodes, for the SC (list=0) or SCL (list=1) program of this decoder. */
int pcd_program_stats(const pcd_decoder* d, int list, unsigned counts[2]);
/* LDS bytes per codeword and codewords (waves) per workgroup the launch will use; 0 if n/a */
int pcd_launch_info(const pcd_decoder* d, int list, unsigned info[4]);
/* Name of the kernel (as rocprofv3 shows it, without namespace and template arguments) that decode calls of this decoder
   launch: list = 0 the SC program's, 1 the list program's.  Which side of a geometry rule a code fell on is visible here:
   "scl_lane_kernel" / "scl_lane_n12_l5_kernel" ... are the lane-per-path kernels, "scl_decode_kernel" the path-group ones,
   "sc_pair_kernel" / "sc_pair_n9_kernel" / "sc_pair_n8_kernel" / "sc_pair_n7_kernel" / "sc_pair_n6_kernel" (N = 1024 / 512 / 256 / 128 / 64) / "sc_decode_kernel" the SC decoders. */
int pcd_kernel_name(const pcd_decoder* d, int list, char* buf, size_t cap);
/* Kernel timing with HIP events recorded on the launch stream around the kernel(s) of every decode call
 * (up to 512 calls after pcd_set_timing(d, 1), which also resets the list).  pcd_kernel_times waits for
 * the recorded events and returns the per-call durations in ms; *count = number of calls recorded. */
int pcd_set_timing(pcd_decoder* d, int enable);
int pcd_kernel_times(pcd_decoder* d, float* ms, size_t cap, size_t* count);

/* =====================================================================================================
 * Batched ENCODER: PolarCode::Encoding::ButterflyFipPacked behind pypolar.PolarEncoder
 *   ButterflyFipPacked(blockLength, frozenBits)   src/polarcode/encoding/butterfly_fip_packed.cpp:27-43
 *   Encoder::encode_vector(void* info, void* code) src/polarcode/encoding/encoder.cpp:79-91
 *   ButterflyFipPacked::encode                    butterfly_fip_packed.cpp:45-59
 *   pypolar.PolarEncoder                          python/bindings/encoder_python.cc:21-61
 * One HIP kernel launch encodes `rows` information words; results are bit-identical to the reference.
 * Status codes, memory-space tags and pcd_last_error() are shared with the decoder.
 * ===================================================================================================== */
typedef struct pce_encoder pce_encoder; /* opaque handle; not thread-safe */

/* frozen[] strictly ascending.  A new encoder is systematic and carries the DUMMY detector
 * (Encoder::Encoder encoder.cpp:17-26) -- unlike a new decoder, which carries CRC-8. */
int pce_create(size_t block_length, const uint32_t* frozen, size_t n_frozen, int device, pce_encoder** out);
void pce_destroy(pce_encoder* e);
int pce_set_error_detection(pce_encoder* e, unsigned size, const char* type); /* same rules as pcd_set_error_detection */
int pce_set_systematic(pce_encoder* e, int systematic);        /* Encoder::setSystematic */
int pce_is_systematic(const pce_encoder* e);
size_t pce_block_length(const pce_encoder* e);                 /* Encoder::blockLength()  */
size_t pce_info_length(const pce_encoder* e);                  /* Encoder::infoLength()   */
size_t pce_info_bytes(const pce_encoder* e);                   /* Encoder::informationByteSize(): (K + 7) / 8 */
size_t pce_code_bytes(const pce_encoder* e);                   /* N / 8 */
size_t pce_frozen_bits(const pce_encoder* e, uint32_t* out, size_t cap);
int pce_error_detection_mode(const pce_encoder* e, char* buf, size_t cap);

/* Batched Encoder::encode_vector.
 *   info : rows x pce_info_bytes() packed information bytes, MSB first (mem = PCD_MEM_HOST | PCD_MEM_DEVICE).
 *          NOT const: like the reference (Detector::generate on the caller's buffer), the checksum field of every
 *          row is overwritten in place with the checksum of the bytes before it.
 *   code : rows x pce_code_bytes() packed code words, code bit i = byte i/8, bit 7 - i%8 (16-byte aligned)
 *   Device pointers: asynchronous on `stream`; host pointers: staged and synchronised before returning. */
int pce_encode(pce_encoder* e, uint8_t* info, size_t rows, uint8_t* code, int mem, void* stream);

/* =====================================================================================================
 * Batched PUNCTURER: PolarCode::Puncturer behind pypolar.Puncturer (rate matching either side of the codec)
 *   Puncturer(blockLength, frozenBitPositions)     src/polarcode/puncturer.cpp:51-65
 *   puncture<T> / depuncture<T>                    include/polarcode/puncturer.h:66-97
 *   puncturePacked                                 src/polarcode/puncturer.cpp:69-89
 *   pypolar.Puncturer                              python/bindings/puncturer_python.cc:21-150
 * The parent block length N is blockLength rounded up to a power of two; the first N - blockLength frozen positions
 * are removed.  One HIP kernel launch moves `rows` rows.
 * ===================================================================================================== */
typedef struct ppu_puncturer ppu_puncturer;
/* PCD_ERR_PUNCTURE if N - block_length exceeds n_frozen (reference: std::out_of_range) */
int ppu_create(size_t block_length, const uint32_t* frozen, size_t n_frozen, int device, ppu_puncturer** out);
void ppu_destroy(ppu_puncturer* u);
size_t ppu_block_length(const ppu_puncturer* u);          /* Puncturer::blockLength()          */
size_t ppu_parent_block_length(const ppu_puncturer* u);   /* Puncturer::parentBlockLength()    */
size_t ppu_output_positions(const ppu_puncturer* u, uint32_t* out, size_t cap); /* blockOutputPositions() */
/* puncture<T>: rows x N elements -> rows x blockLength elements; elem_bytes = 1 (uint8), 4 (float), 8 (double) */
int ppu_puncture(ppu_puncturer* u, const void* in, size_t rows, void* out, unsigned elem_bytes, int mem, void* stream);
/* depuncture<T>: rows x blockLength -> rows x N, punctured positions = 0 */
int ppu_depuncture(ppu_puncturer* u, const void* in, size_t rows, void* out, unsigned elem_bytes, int mem, void* stream);
/* puncturePacked: rows x N/8 bytes -> rows x blockLength/8 bytes, bits MSB first (both lengths multiples of 8) */
int ppu_puncture_packed(ppu_puncturer* u, const uint8_t* in, size_t rows, uint8_t* out, int mem, void* stream);

const char* pcd_last_error(void);
const char* pcd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* POLAR_AMD_H */
