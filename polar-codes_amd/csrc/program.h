// Decode "program": the reference builds a tree of polymorphic node objects per decoder
// (FastSscAvx::createDecoder fastssc_avx_float.cpp:797-896, SclAvx::createDecoder
// scl_avx_float.cpp:624-651) and walks it with virtual calls.  Here the same node-selection
// rules are applied once on the host and flattened into a linear list of 8-byte instructions
// that every wave of the decode kernels interprets in lock-step (the tree depends only on
// (N, frozen set), so control flow is uniform across the whole batch).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pamd {

enum Op : uint8_t {
    OP_END = 0,
    // ---- Fast-SSC (L = 1) ------------------------------------------------------------------
    SC_F = 1,      // llr[s-1] = f(llr[s])                         RateRNode::decode :148-155
    SC_G,          // llr[s-1] = g(llr[s], bits[off .. off+h))
    SC_G0R,        // llr[s-1] = a + b                             ZeroRNode::decode :232-237
    SC_COMBINE,    // bits[off..off+h) ^= bits[off+h..off+2h)      Combine avx_float.h:190-197
    SC_COMBINE0R,  // bits[off..off+h)  = bits[off+h..off+2h)      Combine_0R :199-204
    SC_RONE,       // fused g + hard decision + combine            ROneNode::rightDecode :205-219
    SC_R0,         // leaves, stage = log2(block), off = first bit
    SC_R1,
    SC_REP,
    SC_SPC,
    SC_DREP,
    SC_DSPC,
    SC_DSPC8,
    SC_TREP,
    SC_TYPE5,
    SC_REPONE8,
    SC_ZSPC8,
    SC_ZEROSPC,
    // leaves of the int8 decoder FastSscFipChar (fastssc_fip_char.cpp:219-384): saturating sums in the reference's
    // lane / tree order, unsigned |x| with abs(-128) = 128, flips by char negation (0 and -128 keep their sign)
    SCC_REP,       // RepetitionDecoder / ShortRepetitionDecoder
    SCC_DREP,      // DoubleRepetitionDecoder (n >= 32)
    SCC_SPC,       // SpcDecoder (n > 32) / ShortSpcDecoder
    SCC_ZEROSPC,   // ZeroSpcDecoder (n > 32) / ShortZeroSpcDecoder
    // ---- Fast-SSCL (L > 1) -----------------------------------------------------------------
    SCL_F = 32,    // per path: llr[s-1] = f(llr[s])               RateRNode::decode :229-263
    SCL_G,         // per path: llr[s-1] = g(llr[s], left bits)
    SCL_COMBINE,   // per path: bit[s] = [left ^ right | right]    CombineBitsLong/Short
    SCL_R0,        // RateZeroDecoder::decode :316-337
    SCL_R1,        // RateOneDecoder::decode  :353-413
    SCL_REP,       // RepetitionDecoder::decode :428-481
    SCL_SPC,       // SpcDecoder::decode      :498-621
    // Specialised forms the host substitutes once the launch geometry is known (capi.cpp, specialize_scl):
    // both LLR stages in LDS and fewer than 64 outputs per path, so the kernel can skip every memory-kind
    // decision.  `off` then carries precomputed constants: bits 0-8 woff(stage), 9-17 woff(stage-1),
    // 18-23 the slot-word shift of the stage.
    SCL_F_S = 40,
    SCL_G_S,
    SCL_C_S,       // COMBINE with h < 32 (one bit word per child)
    SCL_C_W,       // COMBINE with h >= 32 (word-wise)
    SCL_SPC_S,     // SPC / R1 leaf of at most 8 LLRs held in LDS, 8 lanes per path (L <= 8 kernels only)
    SCL_R1_S,
    SCL_G0_S,      // F_S, RateZero on its output, G_S with all-zero left bits in one step (all-frozen left child)
    // F / G of stage logN-1 with a virtual input (recomputed from the channel row) for ALL live paths of a codeword from
    // one set of channel loads (every path reads the same row), with the F of the next one or two stages of the left
    // descent applied to the values while they are still in registers (4 lanes per path x 2 codewords, N = 1024).
    // `off`: bit 0 = the op is a G, bits 1-2 = number of fused F levels below (0..2)
    SCL_VTOP,
    // wide F / G (input not virtual, at least 64 outputs per path) with the F of the stage below applied to its
    // outputs while they are in registers: a lane takes the chunks i and i + h/2 of the output block, stores both
    // (the G of the stage below comes back to them) and their f.  Saves one of the two read passes over every arena
    // stage.  `off` bit 0 = the op is a G.
    SCL_WFG,
    // G of stage logN-2 (the right child of a grandchild of the root) with a virtual input as well: the stage logN-2
    // block it reads is recomputed from the channel row through TWO levels (root child, grandchild) for all paths of
    // both codewords at once -- one 16-byte chunk of the 128 outputs per lane (N = 1024, 4 lanes per path x 2 codewords).
    // The matching SCL_VTOP then does not store that block at all (its `off` bit 5), which removes the largest arena
    // stage.  `off`: bit 0 = the root child is the RIGHT one, bit 1 = the grandchild is a RIGHT child, bits 2-3 = fused F
    // levels below (0..1)
    SCL_VG2,
    // ---- lane-per-path list kernel (scl_lane_kernel.hip): 8 paths x 8 codewords per wave, one lane per path -------------
    // Lowered from the plain SCL program by lower_lane (lane_program.cpp).  `stage` = log2 of the block the op works on
    // (the PARENT block for F / G / C), `side` as above.
    LN_VT = 64,    // root + its child in one pass over the channel row: stage-8 block of this path (stage 9 is never stored).
                   // off bit 0: the root's RIGHT child (g with the root's left bits), bit 1: G at stage 9 (else F),
                   // bit 2: the F of stage 8 runs on the values in registers as well (stage 7 goes to LDS)
                   // N != 1024: ALL stages between the channel row and stage 8 in one pass; off bit l: the op at stage logN - l is a G
    LN_F,          // llr[s-1] = f(llr[s])
    LN_G,          // llr[s-1] = g(llr[s] of the path's slot owner, own left bits)
    LN_G0,         // llr[s-1] = a + b: the left child was all-frozen (no fork since the F, left bits zero)
    LN_C,          // bits[s] = [left ^ right | right]
    LN_R0,
    LN_R1,
    LN_REP,
    LN_SPC,
    // Everything between two leaves (or a leaf and an LN_VT) in one instruction: COMBINEs at stages c0 .. c1 up a right
    // spine (the last one writes to `side`, the others are right children), then a G (kind 1) / G0 (kind 2) at stage gs,
    // then Fs from stage f1 down to f0, then an all-frozen leaf of stage r0 (side r0side) if the descent ends at one.
    // off: c0 | c1 << 4 | gs << 8 | kind << 12 | f1 << 16 | f0 << 20 | r0 << 24 | r0side << 28 (0 = absent)
    LN_WALK,
    // N >= 2048: the virtual rows (the path-independent leftmost block of every stage 9 .. logN - 1) only -- emitted in front of a
    // program that does not start with the all-f pass over the channel row (low rates: an all-frozen leaf above stage 8 comes first)
    LN_VROWS,
    // ---- two-lanes-per-codeword SC kernel (sc_pair_kernel.hip): 32 codewords per wave ---------------------------------------
    // Lowered from the plain SC program by lower_pair (pair_program.cpp).  `stage` = LOCAL stage T = log2(block) - 1 (the
    // parent block for F / G / G0R / C / C0R / RONE), `side` = 1 when the finished block is a right child.
    PR_F = 96,
    PR_G,
    PR_G0R,
    PR_C,
    PR_C0R,
    PR_RONE,
    PR_R0,
    PR_R1,
    PR_REP,
    PR_SPC,
    PR_DREP,
    PR_DSPC,
    PR_DSPC8,
    PR_TREP,
    PR_TYPE5,
    PR_REPONE8,
    PR_ZSPC8,
    PR_ZEROSPC,
    PR_ZR1,   // int8 decoder: block of 4 positions, left half frozen, right half free (G0R / RateOne / COMBINE0R of the plain program)
    PR_WALK,  // everything between two leaves, the leaf included (pair_program.cpp)
};

struct Ins {
    uint8_t op;
    uint8_t stage;  // log2 of the block length the op works on (parent block for F/G/COMBINE)
    uint8_t side;   // SCL: 0 = this node is a left child (or the root), 1 = right child
    uint8_t pad;
    uint32_t off;   // first code-bit index of the block (SC bit array); unused by SCL ops
};
static_assert(sizeof(Ins) == 8, "Ins must be 8 bytes");

struct Program {
    std::vector<Ins> ins;
    unsigned leaves = 0, inner = 0;
};

// Returns "" on success, otherwise the text of the std::invalid_argument the reference throws
// (SURVEY Q6).  frozen: strictly ascending indices < N.
// chr: the node selection of the int8 decoders (FastSscFip::createDecoder fastssc_fip_char.cpp:496-570, which never
// throws; SclFip::createDecoder scl_fip_char.cpp:745-775: repetition leaves at every block length)
std::string compile_sc(unsigned N, const std::vector<uint32_t>& frozen, Program& out, bool chr = false);
std::string compile_scl(unsigned N, const std::vector<uint32_t>& frozen, Program& out, bool chr = false);
// Lowers a plain SCL program (compile_scl output: SCL_F / SCL_G / SCL_COMBINE + leaves) of a code with N = 512 .. 4096 to the
// program of the lane-per-path kernel.  Returns false when the tree needs an op that kernel does not have (a leaf above
// stage 8, ...): the caller then keeps the path-group kernels.
// *needs_hi: the tree has an all-frozen leaf above stage 8 (N >= 2048; the *_hi instantiations of the kernel)
bool lower_lane(unsigned logN, const Program& scl, Program& out, bool chr = false, bool* needs_big = nullptr, bool* needs_hi = nullptr);
// Lowers a plain SC program (compile_sc output; chr: of the int8 decoder) of a code with N = 1024 to the program of the
// two-lanes-per-codeword kernel.  Returns false when the tree needs an op that kernel does not have (a double-SPC leaf above 128
// positions, ...): the caller then keeps sc_decode_kernel.  *needs_big: the tree has a leaf above 128 positions (the BIG
// instantiations of the kernel).
bool lower_pair(unsigned logN, const Program& sc, Program& out, bool chr = false, bool* needs_big = nullptr);

// ---- error detection -------------------------------------------------------------------------
// All supported detectors are linear over GF(2) with a zero initial value, so
//   check(bytes) == true  <=>  XOR over the set info bits t of syndrome[t] == 0.
struct Detector {
    unsigned size = 8;       // check bits (0 = dummy)
    bool nr = false;         // 5G NR flavour (type string contained "nr")
    std::string type_name() const;       // Detector::getType()
    unsigned check_bytes() const;        // bytes the checksum field occupies at the end
    // mirrors Detector::check(void*, int bytes) of the selected reference class
    bool check(const uint8_t* data, int bytes) const;
    // syndrome word for info bit t (MSB-first bit t of the packed output), K info bits
    std::vector<uint32_t> syndrome_table(unsigned K) const;

    // ---- encoder side: Detector::generate(void*, int bytes) of the selected reference class -----------
    // writes the checksum of the leading bytes into the checksum field of data[0..bytes)
    void generate(uint8_t* data, int bytes) const;
    // where generate() puts the field for an information word of `bytes` bytes, and how many leading
    // bits it covers (CRC-32 works on whole little-endian words: crc32.cpp:33-41)
    unsigned field_offset(int bytes) const;
    unsigned payload_bits(int bytes) const { return 8u * field_offset(bytes); }
    // generator word for payload bit t: the checksum field it produces on its own, field byte i in bits
    // 8i..8i+7.  generate() == XOR of the words of the set payload bits (all detectors are linear, init 0).
    std::vector<uint32_t> generator_table(int bytes) const;
};

// Tables that place information bits into a code word 32 positions at a time: for word w of the code
// block (bit i of the block = bit i%32 of word i/32), `mask` marks the non-frozen positions, `base` is the
// number of information bits in earlier words, and mv[0..4] are the move masks of the parallel-suffix
// "expand" (bit deposit) of the next popcount(mask) information bits onto the mask.
struct DepositWord {
    uint32_t mask, base, mv[5], pad;
};
static_assert(sizeof(DepositWord) == 32, "DepositWord must be 32 bytes");
std::vector<DepositWord> deposit_table(unsigned N, const std::vector<uint32_t>& frozen);
// reference model of the device-side expand, for self-checks
uint32_t deposit_apply(const DepositWord& d, uint32_t x);

// returns 0 or a pcd_status-compatible negative code; fills det
int make_detector(unsigned size, const std::string& type, Detector& det, std::string& err);

}  // namespace pamd
