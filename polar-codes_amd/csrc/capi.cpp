// C ABI of the MI355X polar decoder (include/polar_amd.h).  Host logic only: argument checking in
// the reference's terms, decode-program compilation, device table upload, launch geometry.
#include "../../include/polar_amd.h"
#ifdef PAMD_DIAG
#include "../../include/polar_amd_diag.h"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <chrono>
#include <cstring>
#include <mutex>
#include <sched.h>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"
#include "kernels.h"
#include "program.h"
#include "scl_lane_geo.h"

// Experiment switches (geometry overrides, fusion stages, ablation, verbose launch geometry) exist only in the
// diagnostic build (`make diag`, -DPAMD_DIAG): the release library reads no environment variable at all.
#ifdef PAMD_DIAG
#define PAMD_ENV(name) std::getenv(name)
#else
#define PAMD_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

unsigned ilog2(size_t n)
{
    unsigned s = 0;
    while (((size_t)1 << s) < n) ++s;
    return s;
}

constexpr size_t kMaxLds = 160 * 1024;  // LDS per CU on gfx950 (a single workgroup may use all of it)
// LDS is handed out in granules; the occupancy API ignores that and over-reports (13 resident single-wave
// workgroups reported for 12 096 B each, 12 measured: the 13th of every CU ran after the others, halving the
// throughput).  1 280 B = 160 KiB / 128 matches every case measured on MI355X.
constexpr size_t kLdsGranule = 1280;
inline size_t lds_blocks_per_cu(size_t bytes)
{
    const size_t rounded = (bytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
    return rounded ? kMaxLds / rounded : 0;
}

}  // namespace

int pamd::set_error(int code, const std::string& msg) { return fail(code, msg); }

struct pcd_decoder {
    unsigned N = 0, logN = 0, L = 1, K = 0, nb = 0;
    int device = 0;
    int dectype = 1;  // 0 char (int8 arithmetic), 1 float, 2 mixed
    bool systematic = true;
    std::vector<uint32_t> frozen;
    pamd::Detector det;
    pamd::Program sc_prog, scl_prog;
    bool has_sc = false, has_scl = false;
    // lane-per-path list kernel (scl_lane_kernel.hip): N = 1024, 4 < L <= 8, float arithmetic, fp32 rows
    pamd::Program lane_prog;
    bool has_lane = false, lane_big = false;  // lane_big: a forking leaf of 256 LLRs (the BIG instantiation of the lane kernel)
    // Small batches (round 4): the kernels above pack 32 (SC) / 8 .. 2 (list) codewords into a wave, and a lone wave takes the
    // same time whether one of its slots is filled or all -- 83 / 217 us per call for one codeword of cfg 2 / cfg 3.  A sibling
    // handle with the shortest lone-wave geometry (ONE codeword per wave: 64 lanes per codeword for SC, 8 lanes per path for lists)
    // takes the calls of up to small_rows rows: 63 / 190 us, flat up to ~1000 rows since every codeword gets a wave of its own.
    pcd_decoder* small = nullptr;
    size_t small_rows = 0;
    bool is_small = false;  // this handle IS such a sibling: no pair / lane kernels, one codeword per wave
    const pamd::LaneKernel* lane_k = nullptr;  // the lane kernel of another geometry (scl_lane_geo.h); nullptr: N = 1024, 8 lanes per codeword
    unsigned lane_cpw = 8;                     // codewords per wave of the lane kernel
    pamd::Ins* d_lane_prog = nullptr;
    pamd::Program pair_prog;          // two-lanes-per-codeword SC kernel (sc_pair_kernel.hip)
    bool has_pair = false, pair_big = false;  // pair_big: the tree has a leaf above 128 positions (the BIG instantiations)
    pamd::Ins* d_pair_prog = nullptr;
    uint32_t* d_ext = nullptr;        // extraction masks per 32 code bits (deposit_table)
    uint32_t* d_syn_bytes = nullptr;  // nb x 256 syndrome words per output byte value

    // launch geometry
    unsigned sc_wave_words = 0, sc_wpb = 4;  // LDS words per codeword, waves per workgroup
    unsigned sc_virtual_top = 0;
    unsigned sc_s0 = 0, sc_arena_cw_words = 0;  // SC: first LLR stage in the arena (logN = none), arena floats per codeword
    int sc_lcl = 4;                           // log2(lanes per codeword): 4 -> four codewords per wave
    int lpl = 3;
    unsigned cwl = 0;  // log2 codewords per wave of the list kernel
    unsigned lmax = 8, path_stride = 0, wppL = 0, wppR = 0, scl_wave_words = 0, scl_wpb = 1;
    unsigned s0 = 0, arena_path_stride = 0, arena_wave_words = 0;
    unsigned sb = 0, arena_bits_off = 0, gbit_left_words = 0, gbit_path_words = 0;
    bool virtual_top = false;
    float* d_arena = nullptr;
    size_t d_arena_cap = 0;

    // device state (created lazily by the first decode call)
    bool dev_ready = false, det_dirty = true;
    int num_cu = 0;
    pamd::Ins* d_sc_prog = nullptr;
    pamd::Ins* d_scl_prog = nullptr;
    uint16_t* d_lut = nullptr;
    uint32_t* d_syn = nullptr;
    // staging for host-pointer calls and the mixed decoder
    void* d_in = nullptr;
    size_t d_in_cap = 0;
    // host-pointer calls: ring of pinned host chunks + device chunks, copy-in / compute streams (decode_host_pipelined)
    static constexpr int kRing = 3;
    void* h_ring[kRing] = { nullptr, nullptr, nullptr };
    void* d_ring[kRing] = { nullptr, nullptr, nullptr };
    size_t ring_bytes = 0;
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    uint8_t* h_res = nullptr;  // pinned staging of the results of a pipelined host call: bytes | verdicts | metrics
    size_t h_res_cap = 0;
    hipEvent_t ev_h2d[kRing] = { nullptr, nullptr, nullptr }, ev_run[kRing] = { nullptr, nullptr, nullptr };
    int8_t* d_i8 = nullptr;  // int8 decoders called with float rows: CharContainer::insertLlr(const float*) on the device
    size_t d_i8_cap = 0;
    uint8_t* d_out = nullptr;
    size_t d_out_cap = 0;
    uint8_t* d_ok = nullptr;
    size_t d_ok_cap = 0;
    float* d_met = nullptr;
    size_t d_met_cap = 0;
    uint32_t* d_list = nullptr;
    size_t d_list_cap = 0;
    uint32_t* d_count = nullptr;
    int sc_blocks_per_cu[2] = { 0, 0 }, scl_blocks_per_cu[2] = { 0, 0 };  // resident workgroups per CU, per kernel variant (fp32 / int8 input)
    // One stream at a time per handle: the scratch (arena, failed-row list, staging) belongs to the handle, so a call on
    // another stream first waits for the event recorded behind the previous call's launches.
    hipEvent_t last_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;

    // kernel timing: a ring of HIP event pairs recorded on the launch stream around the kernel(s) of
    // each decode call; resolved lazily by pcd_kernel_times so the timed calls never synchronise
    bool timing = false;
    std::vector<hipEvent_t> ev;  // 2 * kTimingSlots events
    unsigned ev_count = 0;       // calls recorded since pcd_set_timing(d, 1)
};
constexpr unsigned kTimingSlots = 512;

namespace {

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(PCD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

template <typename T>
int grow(T*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap && p) return PCD_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), bytes ? bytes : 1));
    cap = bytes;
    return PCD_OK;
}

void specialize_scl(pcd_decoder* d);
void fuse_sc(pcd_decoder* d);

void compute_geometry(pcd_decoder* d)
{
    const unsigned N = d->N;
    const unsigned bitw = N >= 32 ? N / 32 : 1;
    // SC: LLR stages below the root + packed bits per codeword; 64/LC codewords per wave.  The stage right below
    // the root (N/2 floats, half of the LLR memory) is not stored but recomputed from the channel row whenever the
    // root's children are inner nodes or leaves reading through the same accessor (always possible; N >= 16).
    d->sc_virtual_top = (N >= 16 && !PAMD_ENV("PCD_SC_NO_VIRTUAL")) ? 1u : 0u;
    d->sc_wave_words = ((d->sc_virtual_top ? N / 2 : N) + bitw + 3) & ~3u;
    // 16 lanes per codeword while four codewords still leave >= 9 waves per CU; 8 lanes (eight codewords per wave share
    // every wave-uniform cost, and the leaves of <= 8 LLRs use all their lanes) up to N = 256 -- measured N = 64 / 128 / 256:
    // 1.80 / 1.17 / 0.77 -> 3.30 / 1.90 / 0.96 G cw/s, but N = 512: 0.43 -> 0.34 (the LDS slices of eight codewords)
    d->sc_lcl = N <= 256 ? 3 : (N <= 2048 ? 4 : 6);
    if (d->dectype == 0 && d->sc_lcl == 3) d->sc_lcl = 4;  // (the int8 reductions need the 32 byte lanes on 16 or 64 lanes)
    if (d->is_small) d->sc_lcl = 6;
    if (const char* e = PAMD_ENV("PCD_SC_LCL")) {  // experiments
        const int v = std::atoi(e);
        d->sc_lcl = (v == 3 || v == 6) ? v : 4;
        if (d->dectype == 0 && d->sc_lcl == 3) d->sc_lcl = 4;  // (the int8 reductions need the 32 byte lanes on 16 or 64 lanes)
    }
    d->sc_s0 = d->logN;
    d->sc_arena_cw_words = 0;
    if ((N >= 2048 || (N >= 512 && PAMD_ENV("PCD_SC_S0"))) && d->sc_virtual_top && !PAMD_ENV("PCD_SC_NO_ARENA") && !PAMD_ENV("PCD_SC_LCL") && !d->is_small) {
        // long blocks: keep 16 lanes per codeword (four codewords per wave) and move the LLR stages from s0 up into a
        // per-codeword arena slice, s0 chosen so that a wave stays within 10 KB of LDS (16 waves per CU)
        unsigned s0 = d->logN - 1;
        while (s0 > 6 && 4u * (((1u << s0) + bitw + 3) & ~3u) * 4u > 10240u) --s0;
        if (const char* e = PAMD_ENV("PCD_SC_S0")) s0 = std::min(d->logN - 1, std::max(6u, (unsigned)std::atoi(e)));
        d->sc_lcl = 4;
        d->sc_s0 = s0;
        d->sc_wave_words = ((1u << s0) + bitw + 3) & ~3u;
        d->sc_arena_cw_words = (N >> 1) - (1u << s0);  // stages s0 .. logN - 2 (stage logN - 1 is recomputed)
    }
    // LDS bank skew: the codewords of a wave make the same accesses at the same offsets of their slices, and a slice
    // stride that is a multiple of 32 words puts them all on the same banks (measured at N = 1024, stride 544: 40 % of
    // the LDS cycles were bank conflicts).  With 16 (8) lanes per codeword a stride of 16 (8) mod 32 words spreads the two
    // (four) codewords of a 32-lane LDS pass over all banks.
    // (Not when the extra words cost a resident wave: N = 4096 sits exactly at 10 KB per wave.)
    if (d->sc_lcl <= 4 && !PAMD_ENV("PCD_SC_NO_SKEW")) {
        const unsigned want = 1u << d->sc_lcl, cws = 64u >> d->sc_lcl;
        const unsigned skewed = d->sc_wave_words + (want + 32u - d->sc_wave_words % 32u) % 32u;
        if (lds_blocks_per_cu((size_t)skewed * 4 * cws) >= lds_blocks_per_cu((size_t)d->sc_wave_words * 4 * cws))
            d->sc_wave_words = skewed;
    }
    if (d->has_sc) fuse_sc(d);
    d->sc_wpb = 1;  // single-wave workgroups: LDS granularity of one wave, nothing is shared between waves anyway
    if (const char* e = PAMD_ENV("PCD_SC_WPB")) d->sc_wpb = (unsigned)std::max(1, std::atoi(e));
    if (d->L > 1) {
        unsigned lprime = 2;  // path groups one codeword needs
        while (lprime < d->L) lprime <<= 1;
        // Several codewords per wave: every codeword follows the same program, so the interpreter's scalar work and
        // the skeleton of the small ops are shared.  L <= 2: 8 lanes per path x 4 codewords; L <= 4: 8 x 2;
        // L <= 8: 4 lanes per path x 2 codewords up to N = 2048 (measured at cfg 3: 12.0 -> 15.4 M cw/s against 8 lanes
        // x 1; from N = 4096 on the wide F / G passes dominate and 8 lanes x 1 codeword is 5-10 % faster).
        // (4 lanes per path for every L <= 8 up to N = 2048 -- L <= 2: 8 codewords per wave, L <= 4: 4, L <= 8: 2; measured against
        //  8 lanes per path x 4 / 2 codewords (profiles/r02_perf_sweep.txt): L = 2 +33 % at N = 128, +6 % at 512, +10 % at 2048,
        //  level at 1024; L = 4 +41 % / +19 % / +10 % / +5 % at N = 128 / 512 / 1024 / 2048; from N = 4096 on the 8-lane forms
        //  are 5-40 % faster again)
        d->cwl = lprime == 2 ? (d->logN <= 11 ? 3u : 2u) : (lprime == 4 ? (d->logN <= 11 ? 2u : 1u) : (lprime == 8 && d->logN <= 11 ? 1u : 0u));
        if (d->is_small) d->cwl = 0;
        if (const char* e = PAMD_ENV("PCD_CWL")) {  // experiments: 0 = one codeword per wave; 1 at L <= 8: 4 lanes per path
            const unsigned v = (unsigned)std::atoi(e);
            d->cwl = (lprime == 2 && v <= 3) || (lprime == 4 && v <= 2) || (lprime <= 8 && v <= 1) ? v : 0u;
            if (lprime == 2 && v == 1) d->cwl = 0;  // (8 lanes x 2 paths x 2 codewords has no kernel)
        }
        const unsigned lmax = lprime << d->cwl;  // path groups (= path buffers) per wave
        d->lmax = lmax;
        d->lpl = 6 - (int)ilog2(lmax);
        const unsigned LP = 64 / lmax;
        const unsigned pad = LP < 32 ? (LP < 4 ? 4 : LP) : 0;  // bank spread; multiple of 4 floats (16-byte LDS vectors)
        auto woff = [](unsigned t) { return t <= 5 ? t : ((1u << (t - 5)) + 4u); };
        // Bit blocks: stage t of a path at word woff(t) of its area (left children + root | right children).
        // Stages >= sb go to the global arena (sb = logN + 1: none); LLR stages [s0, logN) likewise.  Pick (sb, s0) for
        // the most resident waves (up to 16 pay off: measured at cfg 3, s0 = 9 / 8 / 7 / 6 -> 7.5 / 6.3 / 11.1 / 10.8 M
        // cw/s), then the most LLR stages in LDS, then the most bit stages in LDS.
        auto words_L = [&](unsigned sb) { return (sb > d->logN ? woff(d->logN) + bitw : woff(sb)) | 1u; };
        auto words_R = [&](unsigned sb) { return woff(std::min(sb, d->logN)) | 1u; };
        auto lds_words = [&](unsigned s0, unsigned sb) {
            // path LLR buffers | left / right bit areas (rounded to 16 B) | selection scratch: 64 candidate keys at
            // 8 lanes per path, else 64 * (8 / LP) keys + a 32-entry rank table
            const unsigned scratch = LP == 8 ? 64u : 64u * std::max(1u, 8u / LP) + 32u;
            return ((lmax * ((1u << s0) + pad) + lmax * (words_L(sb) + words_R(sb)) + 3) & ~3u) + scratch;
        };
        unsigned best_s0 = d->logN, best_sb = d->logN + 1, best_waves = 0;
        const unsigned lo = d->logN < 6 ? d->logN : 6;
        const unsigned sb_lo = 7;  // every op specialised for LDS works on stages <= 6
        auto waves_of = [&](unsigned s0, unsigned sb) {
            const size_t bytes = (size_t)lds_words(s0, sb) * 4;
            return bytes > kMaxLds ? 0u : (unsigned)std::min<size_t>(16, lds_blocks_per_cu(bytes));
        };
        // all bit blocks in LDS: most waves, then the most LLR stages in LDS
        for (unsigned s0 = d->logN; s0 + 1 > lo; --s0) {
            const unsigned waves = waves_of(s0, d->logN + 1);
            if (waves > best_waves) {
                best_waves = waves;
                best_s0 = s0;
            }
            if (s0 == 0) break;
        }
        // bit blocks of stages >= sb in the arena turn their accesses into flat_* instructions (measured -4 % at
        // cfg 3 with one codeword per wave): only taken when it buys resident waves; then the most LLR stages in
        // LDS, then the most bit stages (cfg 3, two codewords per wave: s0 = 7 / sb = 7 15.9 M cw/s, 6 / 10 15.4 M)
        unsigned hb_s0 = 0, hb_sb = 0, hb_waves = 0;
        for (unsigned s0 = d->logN; s0 + 1 > lo; --s0) {
            for (unsigned sb = d->logN; sb >= sb_lo; --sb) {
                const unsigned waves = waves_of(s0, sb);
                if (waves > hb_waves) {
                    hb_waves = waves;
                    hb_s0 = s0;
                    hb_sb = sb;
                }
            }
            if (s0 == 0) break;
        }
        if (hb_waves > best_waves) {
            best_waves = hb_waves;
            best_s0 = hb_s0;
            best_sb = hb_sb;
        }
        if (const char* dbg = PAMD_ENV("PCD_S0")) best_s0 = std::min(d->logN, std::max(4u, (unsigned)std::atoi(dbg)));  // experiments
        if (const char* dbg = PAMD_ENV("PCD_SB")) best_sb = std::max(sb_lo, (unsigned)std::atoi(dbg));
        if (best_sb > d->logN) best_sb = d->logN + 1;
        d->s0 = best_s0;
        d->sb = best_sb;
        d->wppL = words_L(d->sb);
        d->wppR = words_R(d->sb);
        d->path_stride = (1u << d->s0) + pad;
        d->scl_wave_words = lds_words(d->s0, d->sb);
        d->arena_path_stride = d->s0 < d->logN ? ((1u << d->logN) - (1u << d->s0)) : 0;
        d->arena_bits_off = lmax * d->arena_path_stride;
        d->gbit_left_words = d->sb <= d->logN ? woff(d->logN) + bitw - woff(d->sb) : 0;
        d->gbit_path_words = d->gbit_left_words + (d->sb < d->logN ? woff(d->logN) - woff(d->sb) : 0);
        d->arena_wave_words = (d->arena_bits_off + lmax * d->gbit_path_words + 3u) & ~3u;
        if (!PAMD_ENV("PCD_NO_SPECIALIZE")) specialize_scl(d);
        d->scl_wpb = 1;
    }
}

// Fast-SSC program (flags in Ins::pad / Ins::side, see sc_kernel.hip): a leaf absorbs the F / G / G0R in front of it
// and a right leaf the COMBINE behind it; COMBINEs of consecutive stages (a right spine) run in the instruction that
// finishes the lowest block; when the finished block is a left child, the parent's G runs there as well; the F ops of a
// left descent run in the instruction that starts it.  F / G inputs must be
// plain LDS stages: not the channel row, not the recomputed stage below the root, not an arena stage.
void fuse_sc(pcd_decoder* d)
{
    if (PAMD_ENV("PCD_NO_FUSE")) return;
    auto& v = d->sc_prog.ins;
    auto is_leaf = [](const pamd::Ins& in) { return in.op >= pamd::SC_R0 && in.op <= pamd::SCC_ZEROSPC; };
    const unsigned top = d->logN - (d->sc_virtual_top ? 1u : 0u);  // first stage that is not a stored LDS / arena stage
    auto lds_stage = [&](unsigned t) { return t < top && t < d->sc_s0; };
    std::vector<pamd::Ins> out;
    out.reserve(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        pamd::Ins in = v[i];
        const bool fg = in.op == pamd::SC_F || in.op == pamd::SC_G || in.op == pamd::SC_G0R;
        if (fg && in.pad == 0 && in.side == 0 && in.stage >= 1 && lds_stage(in.stage) && i + 1 < v.size() &&
            is_leaf(v[i + 1]) && v[i + 1].stage + 1u == in.stage && v[i + 1].pad == 0) {
            pamd::Ins leaf = v[i + 1];
            if (leaf.op != pamd::SC_R0)  // (an all-frozen leaf reads nothing: its F / G simply disappears)
                leaf.pad = in.op == pamd::SC_F ? 1 : (in.op == pamd::SC_G ? 2 : 3);
            in = leaf;
            ++i;
        }
        const bool no_c = PAMD_ENV("PCD_NO_FUSE_C") != nullptr;
        // the block this instruction finishes: [coff, coff + 2 ch)
        unsigned coff = in.off, ch = in.stage ? (1u << (in.stage - 1)) : 0u;
        bool finishes = is_leaf(in) || in.op == pamd::SC_COMBINE || in.op == pamd::SC_COMBINE0R || in.op == pamd::SC_RONE;
        bool chain_head = finishes && !is_leaf(in) && in.pad == 0 && !no_c;
        if (is_leaf(in) && (in.pad & 0xFCu) == 0 && i + 1 < v.size() &&
            (v[i + 1].op == pamd::SC_COMBINE || v[i + 1].op == pamd::SC_COMBINE0R) && v[i + 1].stage == in.stage + 1u &&
            v[i + 1].off + (1u << in.stage) == in.off && !no_c) {
            in.pad |= v[i + 1].op == pamd::SC_COMBINE ? 4 : 8;
            coff = v[i + 1].off;
            ch = 1u << in.stage;
            chain_head = true;
            ++i;
        }
        if (chain_head) {
            unsigned k = 0;
            while (k < 7 && i + 1 < v.size() && v[i + 1].op == pamd::SC_COMBINE && v[i + 1].pad == 0 &&
                   (1u << (v[i + 1].stage - 1)) == 2 * ch && v[i + 1].off + 2 * ch == coff) {
                coff = v[i + 1].off;
                ch *= 2;
                ++k;
                ++i;
            }
            in.pad |= (uint8_t)(k << 4);
        }
        // The finished block is a left child: the parent's G comes next and can run right here (unless a leaf is about to
        // absorb it, which also spares the LDS round trip of its output).
        unsigned fstage = 0;  // stage of the F / G / G0R whose left descent may follow inside this instruction
        const bool no_f = PAMD_ENV("PCD_NO_FUSE_F") != nullptr;
        if (finishes && ch && !no_f && !PAMD_ENV("PCD_NO_FUSE_G") && i + 1 < v.size() && v[i + 1].op == pamd::SC_G &&
            v[i + 1].pad == 0 && v[i + 1].side == 0 && (1u << v[i + 1].stage) == 4 * ch && v[i + 1].off == coff &&
            lds_stage(v[i + 1].stage) &&
            !(i + 2 < v.size() && is_leaf(v[i + 2]) && v[i + 2].stage + 1u == v[i + 1].stage)) {
            in.side |= 8;
            fstage = v[i + 1].stage;
            ++i;
        }
        const bool still_fg = in.op == pamd::SC_F || in.op == pamd::SC_G || in.op == pamd::SC_G0R;  // (not absorbed above)
        if (still_fg && in.pad == 0 && in.side == 0 && !no_f) fstage = in.stage;
        if (fstage) {
            unsigned k = 0, st = fstage;
            while (k < 7 && st >= 2 && i + 1 < v.size() && v[i + 1].op == pamd::SC_F && v[i + 1].pad == 0 && v[i + 1].side == 0 &&
                   v[i + 1].stage + 1u == st && lds_stage(st - 1) &&
                   // the follower must not be the F a leaf would absorb (that saves more: the leaf reads its output directly)
                   !(i + 2 < v.size() && is_leaf(v[i + 2]) && v[i + 2].stage + 1u == v[i + 1].stage)) {
                --st;
                ++k;
                ++i;
            }
            in.side |= (uint8_t)k;
        }
        out.push_back(in);
    }
    v.swap(out);
}

// Replace generic inner-node ops by their specialised forms (program.h) for the chosen geometry.
void specialize_scl(pcd_decoder* d)
{
    // Virtual top stage: if the stage below the root would live in the arena and both children of the root are
    // inner nodes with at least 64 outputs per path, it is never stored -- the root's F / G only keep their
    // slot-word bookkeeping (flag 2) and the children's F / G recompute their inputs from the channel row
    // (flag 4, +8 for the right child).  Removes the largest arena stage and its HBM-side traffic.
    if (d->s0 < d->logN && d->logN >= 8 && !PAMD_ENV("PCD_NO_VIRTUAL")) {
        bool ok = !d->scl_prog.ins.empty() && d->scl_prog.ins[0].op == pamd::SCL_F && d->scl_prog.ins[0].stage == d->logN;
        for (const auto& in : d->scl_prog.ins)
            if (in.stage == d->logN - 1 && in.op >= pamd::SCL_R0 && in.op <= pamd::SCL_SPC) ok = false;  // leaf child
        if (ok) {
            bool seen_root_g = false;
            for (auto& in : d->scl_prog.ins) {
                if (in.stage == d->logN && (in.op == pamd::SCL_F || in.op == pamd::SCL_G)) {
                    in.side |= 2;
                    if (in.op == pamd::SCL_G) seen_root_g = true;
                } else if (in.stage == d->logN - 1 && (in.op == pamd::SCL_F || in.op == pamd::SCL_G)) {
                    in.side |= seen_root_g ? (4 | 8) : 4;
                }
            }
            d->virtual_top = true;
        }
    }
    auto woff = [](unsigned t) { return t <= 5 ? t : ((1u << (t - 5)) + 4u); };
    const unsigned fb = 6u - (unsigned)d->lpl - d->cwl;  // bits per slot field: a path buffer of the same codeword
    // The children of the root (virtual input) for all paths of a codeword from one set of channel loads, with the F
    // steps of the left descent below them in the same instruction (SCL_VTOP, scl_kernel.h): N = 1024 with 4 lanes per
    // path x 2 codewords per wave, i.e. one 16-byte chunk of the 256 outputs per lane.
    // (it stores stages logN-2 / logN-3 to the arena and stage logN-4 to LDS, and reads its bit blocks from the arena)
    if (d->virtual_top && d->lpl == 2 && d->cwl == 1 && d->logN == 10 && d->logN * fb <= 32 && d->s0 == d->logN - 3 &&
        d->sb <= d->logN - 2 && !PAMD_ENV("PCD_NO_VTOP")) {
        auto& v = d->scl_prog.ins;
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            pamd::Ins in = v[i];
            if ((in.op == pamd::SCL_F || in.op == pamd::SCL_G) && in.stage == d->logN - 1 && (in.side & 4u)) {
                unsigned k = 0;
                while (k < 2 && i + 1 < v.size() && v[i + 1].op == pamd::SCL_F && v[i + 1].stage + 1u + k == in.stage &&
                       (v[i + 1].side & 0xEu) == 0 && v[i + 1].pad == 0) {
                    ++k;
                    ++i;
                }
                in.off = (in.op == pamd::SCL_G ? 1u : 0u) | (k << 1);
                in.op = pamd::SCL_VTOP;
            }
            out.push_back(in);
        }
        v.swap(out);
    }
    // The G of the root's grandchildren with a virtual input as well (SCL_VG2): its stage logN-2 block is recomputed from
    // the channel row, so the SCL_VTOP in front that produced the block (and consumed it through its fused F) need not
    // store it -- the largest stage leaves the arena.  Same geometry as SCL_VTOP.
    if (d->virtual_top && d->lpl == 2 && d->cwl == 1 && d->logN == 10 && d->logN * fb <= 32 && d->s0 == d->logN - 3 &&
        d->sb <= d->logN - 3 && !PAMD_ENV("PCD_NO_VTOP") && PAMD_ENV("PCD_VG2")) {
        // (measured on MI355X, cfg 3: fabric traffic 25.3x -> 20.1x of the algorithmic bytes, but 27.2 -> 26.4 M cw/s -- the
        //  recomputation costs more than the arena round trip it saves -- so the op stays an experiment of the diagnostic
        //  build: PCD_VG2=1)
        auto& v = d->scl_prog.ins;
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        bool r9 = false, r8 = false;
        long producer = -1;  // index (in out) of the SCL_VTOP whose output block the next stage logN-2 G would read
        for (size_t i = 0; i < v.size(); ++i) {
            pamd::Ins in = v[i];
            if (in.stage == d->logN && in.op == pamd::SCL_G) r9 = true;
            if (in.op == pamd::SCL_VTOP) {
                r8 = (in.off & 1u) != 0;  // the grandchild that follows is the right child iff this op is the G
                producer = ((in.off >> 1) & 3u) >= 1u ? (long)out.size() : -1;  // (its F must be fused: no other reader)
            } else if (in.stage == d->logN - 1) {
                producer = -1;  // (an unfused F / G of the root's child, or its COMBINE: nothing to pair with)
            }
            if (in.op == pamd::SCL_G && in.stage == d->logN - 2 && (in.side & 0xEu) == 0 && in.pad == 0 && producer >= 0) {
                unsigned k = 0;
                if (i + 1 < v.size() && v[i + 1].op == pamd::SCL_F && v[i + 1].stage + 1u == in.stage && (v[i + 1].side & 0xEu) == 0 &&
                    v[i + 1].pad == 0) {
                    k = 1;
                    ++i;
                }
                in.op = pamd::SCL_VG2;
                in.off = (r9 ? 1u : 0u) | (r8 ? 2u : 0u) | (k << 2);
                out[(size_t)producer].off |= 32u;  // the producer's block is never read back: do not store it
                producer = -1;
            }
            out.push_back(in);
        }
        v.swap(out);
    }
    // Wide F / G whose output is read next by the F of the stage below: that F runs on the values in registers
    // (SCL_WFG, scl_kernel.h) -- every stage is then read once (by its G) instead of twice.  The stage below the output
    // must still be a plain stored stage (not specialised away later: this pass runs before F_S / G_S are chosen).
    if (!PAMD_ENV("PCD_NO_WFG")) {
        auto& v = d->scl_prog.ins;
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            pamd::Ins in = v[i];
            const bool plain_fg = (in.op == pamd::SCL_F || in.op == pamd::SCL_G) && (in.side & 0xEu) == 0 && in.pad == 0;
            if (plain_fg && in.stage >= 7) {
                // up to kmax F levels; k levels need chunks that are 32 (h >> k) apart to be whole bit words: h >= 32 << k
                const unsigned kmax = 1u;  // (more levels per instruction measured slower, see scl_kernel.h)
                unsigned k = 0;
                while (k < kmax && i + 1 < v.size() && v[i + 1].op == pamd::SCL_F && v[i + 1].stage + 1u + k == in.stage &&
                       (v[i + 1].side & 0xEu) == 0 && v[i + 1].pad == 0 && (1u << (in.stage - 1)) >= (32u << (k + 1))) {
                    ++k;
                    ++i;
                }
                if (k) {
                    in.off = (in.op == pamd::SCL_G ? 1u : 0u) | (k << 1);
                    in.op = pamd::SCL_WFG;
                }
            }
            out.push_back(in);
        }
        v.swap(out);
    }
    for (auto& in : d->scl_prog.ins) {
        const unsigned s = in.stage;
        if ((in.op == pamd::SCL_SPC || in.op == pamd::SCL_R1) && (d->lpl == 3 || d->lpl == 2) && s <= 3 && s < d->s0 && s < d->logN) {
            in.op = in.op == pamd::SCL_SPC ? pamd::SCL_SPC_S : pamd::SCL_R1_S;
            continue;
        }
        if (in.op != pamd::SCL_F && in.op != pamd::SCL_G && in.op != pamd::SCL_COMBINE) continue;
        // 9-bit word offsets are only read by the ops on small stages (C_S, F_S, G_S: s <= 6); C_W takes just the
        // slot shift -- woff(14) = 516 no longer fits 9 bits and used to spill into the shift field at N = 32768
        const bool small = s <= 6;
        const uint32_t aux = (small ? (woff(s) | (woff(s - 1) << 9)) : 0u) | (((s - 1) * fb) << 18);
        if (in.op == pamd::SCL_COMBINE) {
            in.op = (s - 1 < 5) ? pamd::SCL_C_S : pamd::SCL_C_W;
            in.off = aux;
        } else if (s <= 6 && s < d->s0 && s < d->logN) {
            in.op = in.op == pamd::SCL_F ? pamd::SCL_F_S : pamd::SCL_G_S;
            in.off = aux;
        }
    }
    // Fuse the neighbours of small leaves (4 or 8 lanes per path, blocks of at most 8 LLRs); flags in Ins::pad, see
    // scl_kernel.h.  `F_S leaf` / `G_S leaf`: the leaf computes its input from the parent's LLRs in registers (that
    // stage is read by nobody else).  `F_S R0 G_S leaf`: an all-frozen left sibling only adds to the metric, so it
    // joins the right leaf's prologue.  A right leaf (or a COMBINE) also performs the single-word COMBINEs that
    // follow it directly.  Roughly 40 % of the program's instructions at N = 1024 go away.
    if ((d->lpl == 2 || d->lpl == 3) && !PAMD_ENV("PCD_NO_FUSE")) {
        auto& v = d->scl_prog.ins;
        auto small_leaf = [](const pamd::Ins& in) {
            return in.stage <= 3 && !(in.op == pamd::SCL_REP && in.stage > 2) &&
                   (in.op == pamd::SCL_R0 || in.op == pamd::SCL_REP || in.op == pamd::SCL_SPC_S || in.op == pamd::SCL_R1_S);
        };
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {  // F_S / G_S into the leaf
            const pamd::Ins in = v[i];
            const bool fg = (in.op == pamd::SCL_F_S || in.op == pamd::SCL_G_S) && in.stage <= 4 && (in.side & 0xEu) == 0;
            if (fg && i + 1 < v.size() && small_leaf(v[i + 1]) && v[i + 1].stage + 1u == in.stage) {
                pamd::Ins leaf = v[i + 1];
                leaf.pad = in.op == pamd::SCL_F_S ? 1 : 2;
                out.push_back(leaf);
                ++i;
                continue;
            }
            out.push_back(in);
        }
        v.clear();
        for (size_t i = 0; i < out.size(); ++i) {  // R0 left sibling into the right leaf
            if (i + 1 < out.size() && out[i].op == pamd::SCL_R0 && out[i].pad == 1 && out[i + 1].pad == 2 &&
                out[i + 1].stage == out[i].stage && i + 2 < out.size() && out[i + 2].op == pamd::SCL_C_S &&
                out[i + 2].stage == out[i].stage + 1u && !PAMD_ENV("PCD_NO_FUSE_R0") && !PAMD_ENV("PCD_NO_FUSE_C")) {
                // (the COMBINE right behind the pair is fused below, and knows that the left bits are all zero)
                pamd::Ins leaf = out[i + 1];
                leaf.pad = 3;
                v.push_back(leaf);
                ++i;
                continue;
            }
            v.push_back(out[i]);
        }
        out.clear();
        for (size_t i = 0; i < v.size(); ++i) {  // COMBINE chains
            pamd::Ins head = v[i];
            const bool leaf = small_leaf(head) && d->logN > head.stage;
            if ((leaf || head.op == pamd::SCL_C_S) && !PAMD_ENV("PCD_NO_FUSE_C")) {
                unsigned top = head.stage, k = 0, fside = head.side & 1u;  // stage of the block held so far
                if (head.op == pamd::SCL_C_S) k = 1;
                while (i + 1 < v.size() && v[i + 1].op == pamd::SCL_C_S && v[i + 1].pad == 0 && v[i + 1].stage == top + 1u &&
                       top + 1u <= 5u && k < 7u) {
                    fside = v[i + 1].side & 1u;
                    ++top;
                    ++k;
                    ++i;
                }
                unsigned wide = 0;  // ... and the two-word COMBINE of stage 6 on top of a chain that reached stage 5
                if (top == 5 && k > 0 && i + 1 < v.size() && v[i + 1].op == pamd::SCL_C_W && v[i + 1].stage == 6 &&
                    v[i + 1].pad == 0 && !PAMD_ENV("PCD_NO_FUSE_CW")) {
                    fside = v[i + 1].side & 1u;
                    wide = 1;
                    ++i;
                }
                if (wide || k > (head.op == pamd::SCL_C_S ? 1u : 0u))
                    head.pad |= (uint8_t)((k << 2) | (fside << 5) | (wide << 6));
            }
            out.push_back(head);
        }
        v.swap(out);
    }
    // All-frozen left child of a small inner node: `F_S R0 G_S` (or `R0 G_S` with the F_S already inside the R0) is
    // one step -- the metric takes the sum over min(f, 0) and the right child's input is a + b.
    if (!PAMD_ENV("PCD_NO_FUSE") && !PAMD_ENV("PCD_NO_FUSE_G0")) {
        auto& v = d->scl_prog.ins;
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            if (i + 2 < v.size() && v[i].op == pamd::SCL_F_S && (v[i].side & 0xEu) == 0 && v[i + 1].op == pamd::SCL_R0 &&
                v[i + 1].pad == 0 && v[i + 1].stage + 1u == v[i].stage && v[i + 2].op == pamd::SCL_G_S &&
                v[i + 2].stage == v[i].stage) {
                pamd::Ins g = v[i + 2];
                g.op = pamd::SCL_G0_S;
                out.push_back(g);
                i += 2;
            } else if (i + 1 < v.size() && v[i].op == pamd::SCL_R0 && v[i].pad == 1 && v[i + 1].op == pamd::SCL_G_S &&
                       v[i + 1].stage == v[i].stage + 1u) {
                pamd::Ins g = v[i + 1];
                g.op = pamd::SCL_G0_S;
                out.push_back(g);
                i += 1;
            } else {
                out.push_back(v[i]);
            }
        }
        v.swap(out);
    }
    // `X_S(4) [F_S(3) leaf(2)]`: the leaf's lanes hold exactly the stage-3 pairs its F needs, so the op in front joins too
    if (d->lpl == 2 && !PAMD_ENV("PCD_NO_FUSE") && !PAMD_ENV("PCD_NO_FUSE_GRAND")) {
        auto& v = d->scl_prog.ins;
        std::vector<pamd::Ins> out;
        out.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const pamd::Ins& x = v[i];
            const bool head = (x.op == pamd::SCL_F_S || x.op == pamd::SCL_G_S || x.op == pamd::SCL_G0_S) && x.stage == 4 && (x.side & 0xEu) == 0 && x.pad == 0;
            if (head && i + 1 < v.size() && v[i + 1].stage == 2 && (v[i + 1].pad & 3u) == 1 &&
                (v[i + 1].op == pamd::SCL_R0 || v[i + 1].op == pamd::SCL_REP || v[i + 1].op == pamd::SCL_SPC_S ||
                 v[i + 1].op == pamd::SCL_R1_S)) {
                pamd::Ins leaf = v[i + 1];
                leaf.off = (x.op == pamd::SCL_F_S ? 1u : (x.op == pamd::SCL_G_S ? 2u : 3u)) << 24;
                out.push_back(leaf);
                ++i;
            } else {
                out.push_back(x);
            }
        }
        v.swap(out);
    }
}

int ensure_device(pcd_decoder* d, pamd::DeviceScope& scope)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(PCD_ERR_NO_DEVICE,
                    "no HIP device available: the decoder runs on the GPU only (there is no CPU fallback)");
    if (d->device < 0 || d->device >= count) return fail(PCD_ERR_NO_DEVICE, "invalid HIP device ordinal");
    HIP_TRY(scope.enter(d->device));
    if (!d->dev_ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, d->device));
        d->num_cu = prop.multiProcessorCount;
        if (d->has_sc) {
            const size_t b = d->sc_prog.ins.size() * sizeof(pamd::Ins);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_sc_prog), b));
            HIP_TRY(hipMemcpy(d->d_sc_prog, d->sc_prog.ins.data(), b, hipMemcpyHostToDevice));
        }
        if (d->has_scl) {
            const size_t b = d->scl_prog.ins.size() * sizeof(pamd::Ins);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_scl_prog), b));
            HIP_TRY(hipMemcpy(d->d_scl_prog, d->scl_prog.ins.data(), b, hipMemcpyHostToDevice));
        }
        if (d->has_lane) {
            const size_t b = d->lane_prog.ins.size() * sizeof(pamd::Ins);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_lane_prog), b));
            HIP_TRY(hipMemcpy(d->d_lane_prog, d->lane_prog.ins.data(), b, hipMemcpyHostToDevice));
        }
        if (d->has_pair) {
            const size_t b = d->pair_prog.ins.size() * sizeof(pamd::Ins);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_pair_prog), b));
            HIP_TRY(hipMemcpy(d->d_pair_prog, d->pair_prog.ins.data(), b, hipMemcpyHostToDevice));
        }
        if (d->has_lane || d->has_pair) {
            const std::vector<pamd::DepositWord> ext = pamd::deposit_table(d->N, d->frozen);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_ext), ext.size() * sizeof(pamd::DepositWord)));
            HIP_TRY(hipMemcpy(d->d_ext, ext.data(), ext.size() * sizeof(pamd::DepositWord), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_syn_bytes), (size_t)std::max(1u, d->nb) * 256 * 4));
        }
        // information-bit LUT (BitContainer::calculateLUT bitcontainer.cpp:68-84)
        std::vector<uint16_t> lut;
        lut.reserve(d->K);
        size_t fc = 0;
        for (unsigned i = 0; i < d->N; ++i) {
            if (fc < d->frozen.size() && d->frozen[fc] == i) {
                ++fc;
                continue;
            }
            lut.push_back((uint16_t)i);
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_lut), std::max<size_t>(2, lut.size() * 2)));
        if (!lut.empty()) HIP_TRY(hipMemcpy(d->d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_syn), std::max<size_t>(4, (size_t)d->K * 4)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d->d_count), 4));
        d->dev_ready = true;
        d->det_dirty = true;
    }
    if (d->det_dirty) {
        const std::vector<uint32_t> syn = d->det.syndrome_table(d->K);
        if (!syn.empty()) HIP_TRY(hipMemcpy(d->d_syn, syn.data(), syn.size() * 4, hipMemcpyHostToDevice));
        if ((d->has_lane || d->has_pair) && d->nb) {
            // the same syndrome per (output byte, byte value): information bit 8 j + i is bit 7 - i of byte j
            std::vector<uint32_t> tb((size_t)d->nb * 256, 0u);
            for (unsigned j = 0; j < d->nb; ++j)
                for (unsigned v = 1; v < 256; ++v) {
                    uint32_t x = 0;
                    for (unsigned i = 0; i < 8; ++i)
                        if (((v >> (7 - i)) & 1u) && 8 * j + i < d->K) x ^= syn[8 * j + i];
                    tb[(size_t)j * 256 + v] = x;
                }
            HIP_TRY(hipMemcpy(d->d_syn_bytes, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
        }
        d->det_dirty = false;
    }
    return PCD_OK;
}

pamd::DecodeParams base_params(const pcd_decoder* d, const void* llr, size_t rows, uint8_t* out, uint8_t* ok,
                               float* metrics)
{
    pamd::DecodeParams P;
    std::memset(&P, 0, sizeof(P));
    P.llr = llr;
    P.rowidx = nullptr;
    P.rows = (uint32_t)rows;
    P.N = d->N;
    P.logN = d->logN;
    P.K = d->K;
    P.nb = d->nb;
    P.L = d->L;
    P.lut = d->d_lut;
    P.syn = d->d_syn;
    P.out = out;
    P.ok = ok;
    P.metrics = metrics;
    P.det_on = d->det.size != 0;
    P.chr = d->dectype == 0 ? 1u : 0u;
    P.systematic = d->systematic ? 1u : 0u;
    if (const char* ab = PAMD_ENV("PCD_ABLATE")) P.ablate = (uint32_t)std::atoi(ab);
    return P;
}

// two-lanes-per-codeword SC kernel: 32 codewords per single-wave workgroup, 4 workgroups per CU (40 KB of LDS and all 512
// registers of a SIMD lane each)
int run_sc_pair(pcd_decoder* d, pamd::DecodeParams P, bool i8, hipStream_t st)
{
    P.prog = d->d_pair_prog;
    P.ext = d->d_ext;
    P.syn_bytes = d->d_syn_bytes;
    P.prog_len = (uint32_t)d->pair_prog.ins.size() - 2u;
    P.metrics = nullptr;
    const unsigned need = (unsigned)((P.rows + 31u) / 32u);
    unsigned per_cu = 4;
    if (const char* e = PAMD_ENV("PCD_PAIR_BLOCKS_PER_CU")) per_cu = (unsigned)std::max(1, std::min(4, std::atoi(e)));  // experiments
    unsigned grid = std::max(1u, std::min(need, (unsigned)d->num_cu * per_cu));
    if (const char* e = PAMD_ENV("PCD_PAIR_GRID")) grid = (unsigned)std::max(1, std::atoi(e));  // experiments
    if (d->logN == 9) HIP_TRY(pamd::launch_sc_pair_n9(P, i8, d->pair_big, grid, st));
    else if (d->logN == 8) HIP_TRY(pamd::launch_sc_pair_n8(P, i8, d->pair_big, grid, st));
    else if (d->logN == 7) HIP_TRY(pamd::launch_sc_pair_n7(P, i8, d->pair_big, grid, st));
    else if (d->logN == 6) HIP_TRY(pamd::launch_sc_pair_n6(P, i8, d->pair_big, grid, st));
    else HIP_TRY(pamd::launch_sc_pair(P, i8, d->pair_big, grid, st));
    return PCD_OK;
}

int run_sc(pcd_decoder* d, pamd::DecodeParams P, bool i8, hipStream_t st)
{
    // (fp32 or int8 rows for the float decoder; the int8 decoder sees int8 rows only: float rows were converted by the caller)
    if (d->has_pair && (i8 || d->dectype != 0)) return run_sc_pair(d, P, i8, st);
    P.prog = d->d_sc_prog;
    P.prog_len = (uint32_t)d->sc_prog.ins.size() - 1u;  // (without the closing OP_END)
    P.wave_lds_words = d->sc_wave_words;
    P.metrics = nullptr;
    P.sc_virtual_top = d->sc_virtual_top;
    P.s0 = d->sc_s0;
    P.arena_wave_words = d->sc_arena_cw_words;  // (per codeword for the SC kernel)
    P.prog_root_g0r = (!d->sc_prog.ins.empty() && d->sc_prog.ins[0].op == pamd::SC_G0R) ? 1u : 0u;
    const unsigned cw_per_wave = 64u >> d->sc_lcl;
    const size_t lds = (size_t)d->sc_wpb * cw_per_wave * d->sc_wave_words * 4;
    int& sc_bpc = d->sc_blocks_per_cu[i8 ? 1 : 0];
    if (sc_bpc == 0) {  // resident workgroups per CU: the grid is sized to exactly fill the chip
        int occ = 0;
        pamd::DecodeParams Q = P;
        Q.cycles = nullptr;  // (the query is about the production kernel, whatever this call is)
        HIP_TRY(pamd::launch_sc(Q, d->sc_lcl, i8, 1, d->sc_wpb, lds, st, &occ));
        sc_bpc = std::max(1, std::min(occ, (int)lds_blocks_per_cu(lds)));
        if (const char* e = PAMD_ENV("PCD_SC_BLOCKS_PER_CU")) sc_bpc = std::max(1, std::min(64, std::atoi(e)));  // experiments
        if (PAMD_ENV("PCD_VERBOSE"))
            std::fprintf(stderr, "[pcd] sc: %u codewords/wave, lds=%zu B/workgroup, occupancy=%d blocks/CU\n", cw_per_wave, lds, occ);
    }
    const unsigned cw_per_block = d->sc_wpb * cw_per_wave;
    const unsigned need = (unsigned)((P.rows + cw_per_block - 1) / cw_per_block);
    const unsigned grid = std::max(1u, std::min(need, (unsigned)(d->num_cu * sc_bpc)));
    if (d->sc_arena_cw_words) {
        int rc = grow(d->d_arena, d->d_arena_cap, (size_t)grid * d->sc_wpb * cw_per_wave * d->sc_arena_cw_words * sizeof(float));
        if (rc) return rc;
    }
    P.arena = d->d_arena;
    HIP_TRY(pamd::launch_sc(P, d->sc_lcl, i8, grid, d->sc_wpb, lds, st));
    return PCD_OK;
}

// lane-per-path kernel: 8 codewords per single-wave workgroup, 4 workgroups per CU (40 KB of LDS and all 512 registers
// of a SIMD lane each), one arena slice per workgroup
int run_scl_lane(pcd_decoder* d, pamd::DecodeParams P, hipStream_t st)
{
    P.prog = d->d_lane_prog;
    P.ext = d->d_ext;
    P.syn_bytes = d->d_syn_bytes;
    P.prog_len = (uint32_t)d->lane_prog.ins.size() - 1u;
    const unsigned need = (unsigned)((P.rows + d->lane_cpw - 1u) / d->lane_cpw);
    unsigned per_cu = 4;
    if (const char* e = PAMD_ENV("PCD_LANE_BLOCKS_PER_CU")) per_cu = (unsigned)std::max(1, std::min(4, std::atoi(e)));  // experiments
    const unsigned grid = std::max(1u, std::min(need, (unsigned)d->num_cu * per_cu));
    const unsigned arena_words = d->lane_k ? d->lane_k->arena_words() : pamd::scl_lane_arena_words();
    int rc = grow(d->d_arena, d->d_arena_cap, (size_t)grid * arena_words * sizeof(uint32_t));
    if (rc) return rc;
    P.arena = d->d_arena;
    const bool diag_k = P.trace != nullptr || P.cycles != nullptr || P.ablate != 0;
    if (d->lane_k) HIP_TRY(d->lane_k->launch(P, diag_k, grid, st));
    else if (!P.chr && P.i8rows) HIP_TRY((d->lane_big ? pamd::launch_scl_lane_i8_big : pamd::launch_scl_lane_i8)(P, diag_k, grid, st));
    else if (P.chr && d->lane_big) HIP_TRY(pamd::launch_scl_lane_char_big(P, diag_k, grid, st));
    else if (P.chr) HIP_TRY(pamd::launch_scl_lane_char(P, diag_k, grid, st));
    else if (d->lane_big) HIP_TRY(pamd::launch_scl_lane_big(P, diag_k, grid, st));
    else HIP_TRY(pamd::launch_scl_lane(P, diag_k, grid, st));
    return PCD_OK;
}

int run_scl(pcd_decoder* d, pamd::DecodeParams P, bool i8, hipStream_t st)
{
    // The lane-per-path kernels read the rows as they are: fp32, or int8 widened in the top-of-tree pass (round 4: no widened copy
    // of the batch any more).  The int8 decoder's kernels read int8 rows only (float rows were converted by the caller), the float
    // decoder's have an int8-row twin in the default geometry; elsewhere int8 rows into the float decoder keep the path groups.
    if (d->has_lane && (!i8 || d->dectype == 0 || !d->lane_k)) {
        P.i8rows = i8 ? 1u : 0u;
        return run_scl_lane(d, P, st);
    }
    P.prog = d->d_scl_prog;
    if (!P.prog_len) P.prog_len = (uint32_t)d->scl_prog.ins.size() - 1u;  // (without the closing OP_END)
    P.wave_lds_words = d->scl_wave_words;
    P.path_stride = d->path_stride;
    P.wppL = d->wppL;
    P.wppR = d->wppR;
    P.s0 = d->s0;
    P.arena_path_stride = d->arena_path_stride;
    P.arena_wave_words = d->arena_wave_words;
    P.arena_bits_off = d->arena_bits_off;
    P.cwl = d->cwl;
    P.sb = d->sb;
    P.gbit_left_words = d->gbit_left_words;
    P.gbit_path_words = d->gbit_path_words;
    const size_t lds = (size_t)d->scl_wpb * d->scl_wave_words * 4;
    int& scl_bpc = d->scl_blocks_per_cu[i8 ? 1 : 0];
    if (scl_bpc == 0) {  // resident workgroups per CU (LDS and VGPR limited)
        int occ = 0;
        pamd::DecodeParams Q = P;
        Q.trace = nullptr;
        Q.cycles = nullptr;
        Q.ablate = 0;
        HIP_TRY(pamd::launch_scl(Q, d->lpl, i8, 1, d->scl_wpb, lds, st, &occ));
        scl_bpc = std::max(1, std::min(occ, (int)lds_blocks_per_cu(lds)));
        if (const char* e = PAMD_ENV("PCD_BLOCKS_PER_CU")) scl_bpc = std::max(1, std::min(64, std::atoi(e)));  // experiments
        if (PAMD_ENV("PCD_VERBOSE")) std::fprintf(stderr, "[pcd] scl: %u lanes/path x %u codewords/wave, s0=%u sb=%u lds=%zu B occupancy=%d -> %d blocks/CU\n", 64u / d->lmax, 1u << d->cwl, d->s0, d->sb, lds, occ, scl_bpc);
    }
    const unsigned cw_per_block = d->scl_wpb << d->cwl;
    const unsigned need = (unsigned)((P.rows + cw_per_block - 1) / cw_per_block);
    const unsigned grid = std::max(1u, std::min(need, (unsigned)(d->num_cu * scl_bpc)));
    if (d->arena_wave_words) {
        int rc = grow(d->d_arena, d->d_arena_cap, (size_t)grid * d->scl_wpb * d->arena_wave_words * sizeof(float));
        if (rc) return rc;
    }
    P.arena = d->d_arena;
    HIP_TRY(pamd::launch_scl(P, d->lpl, i8, grid, d->scl_wpb, lds, st));
    return PCD_OK;
}

// all pointers are device pointers here
int decode_device(pcd_decoder* d, const void* llr, bool i8, size_t rows, uint8_t* out, uint8_t* ok,
                  float* metrics, hipStream_t st)
{
    if (rows == 0) return PCD_OK;
    if (rows > 0xFFFFFFFFull) return fail(PCD_ERR_INVALID_ARG, "rows must fit in 32 bits");
    int rc;
    if (d->has_last && d->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, d->last_ev, 0));
    const bool timed = d->timing && d->ev_count < kTimingSlots;
    if (timed) {
        if (d->ev.empty()) {
            d->ev.resize(2 * kTimingSlots, nullptr);
            for (auto& e : d->ev) HIP_TRY(hipEventCreate(&e));
        }
        HIP_TRY(hipEventRecord(d->ev[2 * d->ev_count], st));
    }
    if (d->dectype == 0 && !i8) {
        // float rows into an int8 decoder: CharContainer::insertLlr(const float*) bitcontainer.cpp:505-516 -- round to
        // nearest even, saturate to [-128, 127] (convert_f32_to_int8_large / vectorizedFtoC; N >= 8 here)
        if ((rc = grow(d->d_i8, d->d_i8_cap, rows * d->N))) return rc;
        HIP_TRY(pamd::launch_f32_to_i8(reinterpret_cast<const float*>(llr), d->d_i8, rows * d->N, st));
        llr = d->d_i8;
        i8 = true;
    }
    if (d->L <= 1) {
        if (metrics) HIP_TRY(hipMemsetAsync(metrics, 0, rows * sizeof(float), st));
        if ((rc = run_sc(d, base_params(d, llr, rows, out, ok, nullptr), i8, st))) return rc;
    } else if (d->dectype == 2) {
        // AdaptiveFloat::decode adaptive_float.cpp:33-45: SC for every row, SCL only where the
        // detector rejected the SC result
        uint8_t* okbuf = ok;
        if (!okbuf) {
            if ((rc = grow(d->d_ok, d->d_ok_cap, rows))) return rc;
            okbuf = d->d_ok;
        }
        if (metrics) HIP_TRY(hipMemsetAsync(metrics, 0, rows * d->L * sizeof(float), st));
        if ((rc = run_sc(d, base_params(d, llr, rows, out, okbuf, nullptr), i8, st))) return rc;
        if ((rc = grow(d->d_list, d->d_list_cap, rows * 4))) return rc;
        HIP_TRY(hipMemsetAsync(d->d_count, 0, 4, st));
        HIP_TRY(pamd::launch_collect_failed(okbuf, (uint32_t)rows, d->d_list, d->d_count, st));
        {
            // the list pass reads its row count from the device (no host synchronisation): the grid is sized for the
            // worst case, waves without work leave at once
            pamd::DecodeParams P = base_params(d, llr, rows, out, okbuf, metrics);
            P.rowidx = d->d_list;
            P.rows_dev = d->d_count;
            if ((rc = run_scl(d, P, i8, st))) return rc;
        }
    } else {
        if ((rc = run_scl(d, base_params(d, llr, rows, out, ok, metrics), i8, st))) return rc;
    }
    if (timed) {
        HIP_TRY(hipEventRecord(d->ev[2 * d->ev_count + 1], st));
        d->ev_count++;
    }
    if (!d->last_ev) HIP_TRY(hipEventCreateWithFlags(&d->last_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(d->last_ev, st));
    d->last_stream = st;
    d->has_last = true;
    return PCD_OK;
}

// Host pointers, large batches: the rows are cut into chunks that travel through a ring of PINNED host buffers.  A few
// host threads copy chunk k + 1 .. k + 2 from the caller's (pageable) memory into the ring while the DMA engine moves
// chunk k to the device and the decode kernel works on chunk k - 1 -- a copy from pageable memory would be staged by the
// runtime synchronously and serialise with the kernel.  The results of a chunk (65 bytes per row, + metrics) follow its kernel
// on a third stream into a pinned staging buffer; at the end the same threads copy it into the caller's (pageable) arrays,
// a slice each.
// Returns PCD_OK, an error, or kPipelineUnavailable when the worker threads could not be started (the caller then takes
// the simple staged path).
constexpr int kPipelineUnavailable = 1;
int decode_host_pipelined(pcd_decoder* d, const void* llr, bool i8, size_t rows, float* dmet, uint8_t* out, uint8_t* ok, float* metrics,
                          bool* results_delivered)
{
    constexpr int R = pcd_decoder::kRing;
    const size_t row_bytes = (size_t)d->N * (i8 ? 1 : 4);
    // chunks of at most 16 Mi LLRs -- 64 MiB of fp32 rows, 16 MiB of int8 rows: sized by ROWS, what a chunk costs is its
    // kernel time, and the first chunk's copies are the pipeline's fill time -- (the ring pins three of them on the host and
    // holds three on the device), whole rows, a multiple of 16 rows so that every chunk starts 16-byte aligned
    size_t chunk_llrs = (size_t)16 << 20;
    if (const char* e = PAMD_ENV("PCD_CHUNK_LLRS")) chunk_llrs = (size_t)std::max(1, std::atoi(e)) << 20;  // experiments (Mi LLRs)
    size_t chunk_rows = std::max<size_t>(1, chunk_llrs / d->N);
    if (chunk_rows > 16) chunk_rows &= ~(size_t)15;
    const size_t chunk_bytes = chunk_rows * row_bytes;
    const size_t nchunks = (rows + chunk_rows - 1) / chunk_rows;
    int rc;
    if (d->ring_bytes < chunk_bytes) {
        for (int b = 0; b < R; ++b) {
            if (d->h_ring[b]) HIP_TRY(hipHostFree(d->h_ring[b]));
            if (d->d_ring[b]) HIP_TRY(hipFree(d->d_ring[b]));
            d->h_ring[b] = d->d_ring[b] = nullptr;
        }
        d->ring_bytes = 0;
        for (int b = 0; b < R; ++b) {
            HIP_TRY(hipHostMalloc(&d->h_ring[b], chunk_bytes, hipHostMallocDefault));
            HIP_TRY(hipMalloc(&d->d_ring[b], chunk_bytes));
        }
        d->ring_bytes = chunk_bytes;
    }
    const unsigned Lm = std::max(1u, d->L);
    const size_t out_bytes = rows * d->nb, met_bytes = dmet ? rows * Lm * sizeof(float) : 0;
    const size_t res_bytes = out_bytes + rows + met_bytes + 64;
    // int8 rows: the kernel is the bound and the results (a quarter of the input bytes) are worth hiding behind it; fp32 rows: the
    // link is, every further copy in flight costs it something, and the caller fetches the results at the end (measured:
    // 21.1 -> 22.0 M cw/s for int8 rows, 12.1 -> 11.8 M for fp32 rows with the results streamed)
    const bool stream_results = i8;
    *results_delivered = stream_results;
    if (stream_results && d->h_res_cap < res_bytes) {
        if (d->h_res) HIP_TRY(hipHostFree(d->h_res));
        d->h_res = nullptr;
        d->h_res_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d->h_res), res_bytes, hipHostMallocDefault));
        d->h_res_cap = res_bytes;
    }
    uint8_t* const h_out = d->h_res;
    // (the staging buffer exists only when the results are streamed: no pointer arithmetic on a null base otherwise)
    uint8_t* const h_ok = stream_results ? d->h_res + out_bytes : nullptr;
    float* const h_met = stream_results ? reinterpret_cast<float*>(d->h_res + ((out_bytes + rows + 15) & ~(size_t)15)) : nullptr;
    if (!d->s_out) HIP_TRY(hipStreamCreateWithFlags(&d->s_out, hipStreamNonBlocking));
    if (!d->s_in) {
        HIP_TRY(hipStreamCreateWithFlags(&d->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&d->s_run, hipStreamNonBlocking));
        for (int b = 0; b < R; ++b) {
            HIP_TRY(hipEventCreateWithFlags(&d->ev_h2d[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d->ev_run[b], hipEventDisableTiming));
        }
    }
    // worker threads: thread t copies slice t of every chunk once the main thread has released the chunk's ring slot.
    // As many as the process may run on (its affinity mask, not the machine), at most 16; they sleep on a condition
    // variable between chunks.
    unsigned hw = 1;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = (unsigned)std::max(1, CPU_COUNT(&set));
    }
    const unsigned T = std::min(16u, std::max(2u, hw / 2));
    std::mutex mu;
    std::condition_variable cv_rel, cv_done, cv_res;
    int res_state = 0;                      // 0: results not there yet, 1: copy them out, 2: give up   (guarded by mu)
    long released = -1;                     // chunks 0 .. released may be copied      (guarded by mu)
    std::vector<unsigned> done(nchunks, 0);  // threads that have copied their slice   (guarded by mu)
    const uint8_t* src = reinterpret_cast<const uint8_t*>(llr);
    auto worker = [&](unsigned t) {
        for (size_t k = 0; k < nchunks; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_rel.wait(lk, [&] { return released >= (long)k; });
                if (released >= (long)nchunks) return;  // (shut down)
            }
            const size_t r0 = k * chunk_rows, nr = std::min(chunk_rows, rows - r0), bytes = nr * row_bytes;
            const size_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
            std::memcpy(reinterpret_cast<uint8_t*>(d->h_ring[k % R]) + lo, src + r0 * row_bytes + lo, hi - lo);
            {
                std::lock_guard<std::mutex> lk(mu);
                ++done[k];
            }
            cv_done.notify_one();
        }
        // the results: slice t of the pinned staging buffer into the caller's arrays
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_res.wait(lk, [&] { return res_state != 0; });
            if (res_state != 1 || !stream_results) return;
        }
        auto slice = [&](void* dst, const void* from, size_t bytes) {
            const size_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
            if (dst && hi > lo) std::memcpy(reinterpret_cast<uint8_t*>(dst) + lo, reinterpret_cast<const uint8_t*>(from) + lo, hi - lo);
        };
        slice(out, h_out, out_bytes);
        slice(ok, h_ok, rows);
        slice(metrics, h_met, met_bytes);
    };
    std::vector<std::thread> pool;
    auto finish = [&](int code) {
        {
            std::lock_guard<std::mutex> lk(mu);
            released = (long)nchunks;  // let the workers run out
            if (res_state == 0) res_state = code == PCD_OK ? 1 : 2;
        }
        cv_rel.notify_all();
        cv_res.notify_all();
        for (auto& th : pool) th.join();
        return code;
    };
    try {
        pool.reserve(T);
        for (unsigned t = 0; t < T; ++t) pool.emplace_back(worker, t);
    } catch (const std::exception&) {  // (thread limit of the container: no pipeline, but no crash either)
        return finish(kPipelineUnavailable);
    }
    size_t rel = 0;  // chunks released to the workers so far
    for (size_t k = 0; k < nchunks; ++k) {
        const int b = (int)(k % R);
        const size_t r0 = k * chunk_rows, nr = std::min(chunk_rows, rows - r0);
        // the workers fill the ring up to R - 1 chunks ahead: a slot is free once the H2D of the chunk that used it
        // last (R chunks earlier, enqueued in an earlier iteration) has completed
        while (rel < nchunks && rel < k + (size_t)R) {
            if (rel >= (size_t)R && hipEventSynchronize(d->ev_h2d[rel % R]) != hipSuccess)
                return finish(fail(PCD_ERR_HIP, "hipEventSynchronize failed in the host-pointer pipeline"));
            {
                std::lock_guard<std::mutex> lk(mu);
                released = (long)rel;
            }
            cv_rel.notify_all();
            ++rel;
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return done[k] >= T; });
        }
        hipError_t e = hipSuccess;
        if (k >= (size_t)R) e = hipStreamWaitEvent(d->s_in, d->ev_run[b], 0);  // device slot b: its kernel has finished
        if (e == hipSuccess) e = hipMemcpyAsync(d->d_ring[b], d->h_ring[b], nr * row_bytes, hipMemcpyHostToDevice, d->s_in);
        if (e == hipSuccess) e = hipEventRecord(d->ev_h2d[b], d->s_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(d->s_run, d->ev_h2d[b], 0);
        if (e != hipSuccess) return finish(fail(PCD_ERR_HIP, std::string("host-pointer pipeline: ") + hipGetErrorString(e)));
        if ((rc = decode_device(d, d->d_ring[b], i8, nr, d->d_out + r0 * d->nb, d->d_ok + r0,
                                dmet ? dmet + r0 * Lm : nullptr, d->s_run)))
            return finish(rc);
        if (hipEventRecord(d->ev_run[b], d->s_run) != hipSuccess) return finish(fail(PCD_ERR_HIP, "hipEventRecord failed"));
        if (!stream_results) continue;
        // the chunk's results behind its kernel, on their own stream
        e = hipStreamWaitEvent(d->s_out, d->ev_run[b], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(h_out + r0 * d->nb, d->d_out + r0 * d->nb, nr * d->nb, hipMemcpyDeviceToHost, d->s_out);
        if (e == hipSuccess) e = hipMemcpyAsync(h_ok + r0, d->d_ok + r0, nr, hipMemcpyDeviceToHost, d->s_out);
        if (e == hipSuccess && dmet) e = hipMemcpyAsync(h_met + r0 * Lm, dmet + r0 * Lm, nr * Lm * sizeof(float), hipMemcpyDeviceToHost, d->s_out);
        if (e != hipSuccess) return finish(fail(PCD_ERR_HIP, std::string("host-pointer pipeline (results): ") + hipGetErrorString(e)));
    }
    {
        const hipError_t e1 = hipStreamSynchronize(d->s_run), e2 = hipStreamSynchronize(d->s_out);
        if (e1 != hipSuccess || e2 != hipSuccess)
            return finish(fail(PCD_ERR_HIP, std::string("host-pointer pipeline: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2)));
    }
    return finish(PCD_OK);  // (the workers copy the results out, then end)
}

int decode_any(pcd_decoder* d, const void* llr, bool i8, size_t rows, uint8_t* out, uint8_t* ok, float* metrics,
               int mem, void* stream)
{
    if (!d) return fail(PCD_ERR_INVALID_ARG, "null decoder");
    // small batches: the sibling with one codeword per wave (not while kernel times are being recorded on this handle: the
    // measurement is about the main kernels)
    if (d->small && rows && rows <= d->small_rows && !d->timing) return decode_any(d->small, llr, i8, rows, out, ok, metrics, mem, stream);
    if (rows && (!llr || !out)) return fail(PCD_ERR_INVALID_ARG, "null llr/out pointer");
    if (d->det.size != 0 && d->nb < d->det.check_bytes())
        return fail(PCD_ERR_INVALID_ARG, "information word shorter than the detector's checksum field");
    if (d->det.size == 32 && d->nb < 4) return fail(PCD_ERR_INVALID_ARG, "CRC-32 needs at least 4 output bytes");
    pamd::DeviceScope scope;
    int rc = ensure_device(d, scope);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mem == PCD_MEM_DEVICE) return decode_device(d, llr, i8, rows, out, ok, metrics, st);
    if (mem != PCD_MEM_HOST) return fail(PCD_ERR_INVALID_ARG, "mem must be 0 (host pointers) or 1 (device pointers)");
    // host pointers: stage through device buffers, synchronous
    const size_t esz = i8 ? 1 : 4;
    const size_t in_bytes = rows * d->N * esz, out_bytes = rows * d->nb;
    if (in_bytes >= ((size_t)96 << 20) && d->dectype != 2) {
        // large batch: pinned ring, copies overlapped with the kernels ("mixed" keeps the simple path: its list pass
        // addresses rows of the whole batch through an index list)
        if ((rc = grow(d->d_out, d->d_out_cap, out_bytes))) return rc;
        if ((rc = grow(d->d_ok, d->d_ok_cap, rows))) return rc;
        float* dm = nullptr;
        if (metrics) {
            if ((rc = grow(d->d_met, d->d_met_cap, rows * std::max(1u, d->L) * sizeof(float)))) return rc;
            dm = d->d_met;
        }
        const auto t0 = std::chrono::steady_clock::now();
        bool delivered = false;
        rc = decode_host_pipelined(d, llr, i8, rows, dm, out, ok, metrics, &delivered);
        if (rc == PCD_OK) {
            if (!delivered) {
                HIP_TRY(hipMemcpy(out, d->d_out, out_bytes, hipMemcpyDeviceToHost));
                if (ok) HIP_TRY(hipMemcpy(ok, d->d_ok, rows, hipMemcpyDeviceToHost));
                if (metrics) HIP_TRY(hipMemcpy(metrics, dm, rows * std::max(1u, d->L) * sizeof(float), hipMemcpyDeviceToHost));
            }
            if (PAMD_ENV("PCD_VERBOSE"))
                std::fprintf(stderr, "[pcd] host pipeline: %.2f ms (rows in, kernels, %zu B of results out)\n",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), out_bytes);
            return PCD_OK;
        }
        if (rc != kPipelineUnavailable) return rc;
        // (no worker threads to be had: the simple staged path below)
    }
    if ((rc = grow(d->d_in, d->d_in_cap, in_bytes))) return rc;
    if ((rc = grow(d->d_out, d->d_out_cap, out_bytes))) return rc;
    if ((rc = grow(d->d_ok, d->d_ok_cap, rows))) return rc;
    float* dmet = nullptr;
    if (metrics) {
        if ((rc = grow(d->d_met, d->d_met_cap, rows * std::max(1u, d->L) * sizeof(float)))) return rc;
        dmet = d->d_met;
    }
    HIP_TRY(hipMemcpyAsync(d->d_in, llr, in_bytes, hipMemcpyHostToDevice, st));
    // the mixed path may use d_ok as its own scratch, so hand it a distinct verdict buffer
    if ((rc = decode_device(d, d->d_in, i8, rows, d->d_out, d->d_ok, dmet, st))) return rc;
    HIP_TRY(hipMemcpyAsync(out, d->d_out, out_bytes, hipMemcpyDeviceToHost, st));
    if (ok) HIP_TRY(hipMemcpyAsync(ok, d->d_ok, rows, hipMemcpyDeviceToHost, st));
    if (metrics) HIP_TRY(hipMemcpyAsync(metrics, dmet, rows * std::max(1u, d->L) * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PCD_OK;
}

}  // namespace

extern "C" {

static int create_decoder(size_t block_length, size_t list_size, const uint32_t* frozen, size_t n_frozen, const char* type, int device,
                          bool small, pcd_decoder** out);

int pcd_create(size_t block_length, size_t list_size, const uint32_t* frozen, size_t n_frozen, const char* type,
               int device, pcd_decoder** out)
{
    const int rc = create_decoder(block_length, list_size, frozen, n_frozen, type, device, false, out);
    if (rc != PCD_OK) return rc;
    pcd_decoder* d = *out;
    // the sibling for small batches (see pcd_decoder::small): only where the main geometry packs codewords into waves
    // (not for the SC decoders of N <= 256: a lone wave of sc_pair_n8 / n7_kernel finishes its up to 32 codewords as fast as the sibling one --
    //  tools/latency_small_n.py, us per call from device pointers, pair / sibling: N = 128 29 / 30, N = 256 36 / 37, N = 512 52 / 48)
    const bool pair_is_quick = d->L <= 1 && d->has_pair && d->logN <= 8;
    // (nor for the list decoders on the 2 / 4-lane kernels and at N <= 256: L = 2, one row per call, lone lane wave / sibling: N = 128 48 / 58,
    //  N = 256 73 / 89, N = 512 123 / 133, N = 1024 213 / 222 us; L = 8: N = 128 49 / 49, N = 256 72 / 73, N = 512 123 / 115, N = 1024 214 / 192)
    const bool lane_is_quick = d->L > 1 && d->dectype != 2 && d->has_lane && d->lane_k && (d->lane_k->lcl <= 2 || d->logN <= 8);
    if ((d->has_pair || d->has_lane) && !pair_is_quick && !lane_is_quick && !PAMD_ENV("PCD_NO_SMALL")) {
        pcd_decoder* sm = nullptr;
        if (create_decoder(block_length, list_size, frozen, n_frozen, type, device, true, &sm) == PCD_OK) {
            d->small = sm;
            // (measured on MI355X, device pointers, us per call: SC 63 against 83 up to 2 048 rows, 78 / 84 there; list decoder L = 8
            //  192 against 217 at one row, 216 / 226 at 1 024, 234 / 231 at 2 048 -- profiles/r04_host_latency.txt)
            d->small_rows = d->L <= 1 ? 2048 : (d->dectype == 2 ? 512 : 1024);
        }
    }
    return PCD_OK;
}

static int create_decoder(size_t block_length, size_t list_size, const uint32_t* frozen, size_t n_frozen, const char* type, int device,
                          bool small, pcd_decoder** out)
{
    if (!out) return fail(PCD_ERR_INVALID_ARG, "null output handle");
    *out = nullptr;
    if (!type) return fail(PCD_ERR_INVALID_ARG, "null decoder type");
    // Decoding::create decoder.cpp:26-52
    std::string t = type;
    std::transform(t.begin(), t.end(), t.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    int flag;
    if (t.find("char") != std::string::npos)
        flag = 0;
    else if (t.find("float") != std::string::npos)
        flag = 1;
    else if (t.find("mixed") != std::string::npos)
        flag = 2;
    else if (t.find("scan") != std::string::npos)
        flag = 3;
    else
        return fail(PCD_ERR_UNKNOWN_TYPE, "Unknown PolarDecoder type!");
    if (list_size < 2 && flag != 0) flag = 1;
    if (flag == 3) return fail(PCD_ERR_UNSUPPORTED, "the SCAN decoder is not on the accelerated path");

    const size_t N = block_length;
    if (N < 8 || (N & (N - 1)) != 0 || N > 32768)
        return fail(PCD_ERR_INVALID_ARG, "blockLength must be a power of two in [8, 32768]");
    if (list_size == 0) return fail(PCD_ERR_INVALID_ARG, "listSize must be >= 1");
    if (list_size > 32) return fail(PCD_ERR_CAPACITY, "listSize > 32 does not fit the 64 lanes of a wavefront");
    if (n_frozen > N) return fail(PCD_ERR_INVALID_ARG, "more frozen positions than code bits");
    if (n_frozen && !frozen) return fail(PCD_ERR_INVALID_ARG, "null frozen set");
    for (size_t i = 0; i < n_frozen; ++i) {
        if (frozen[i] >= N) return fail(PCD_ERR_INVALID_ARG, "frozen position out of range");
        if (i && frozen[i] <= frozen[i - 1])
            return fail(PCD_ERR_INVALID_ARG, "frozen positions must be strictly ascending");
    }

    pcd_decoder* d = new pcd_decoder();
    d->is_small = small;
    d->N = (unsigned)N;
    d->logN = ilog2(N);
    d->L = (unsigned)list_size;
    d->K = (unsigned)(N - n_frozen);
    d->nb = (d->K + 7) / 8;
    d->device = device;
    d->dectype = flag;
    d->frozen.assign(frozen, frozen + n_frozen);
    d->det.size = 8;  // makeDecoder: dec->setErrorDetection(new CRC8()) decoder.cpp:85
    d->det.nr = false;

    const bool chr = d->dectype == 0;  // makeDecoder decoder.cpp:65-67,80-82: FastSscFipChar / SclFipChar
    if (d->L == 1 || d->dectype == 2) {
        const std::string e = pamd::compile_sc(d->N, d->frozen, d->sc_prog, chr);
        if (!e.empty()) {
            delete d;
            return fail(PCD_ERR_FROZEN_SET, e);
        }
        d->has_sc = true;
        // the two-lanes-per-codeword kernel takes the float codes it has every op for (lowered from the plain program,
        // before fuse_sc rewrites it for sc_decode_kernel)
        if (!PAMD_ENV("PCD_NO_PAIR") && !small) d->has_pair = pamd::lower_pair(d->logN, d->sc_prog, d->pair_prog, chr, &d->pair_big);
    }
    if (d->L > 1) {
        pamd::compile_scl(d->N, d->frozen, d->scl_prog, chr);
        d->has_scl = true;
        // the lane-per-path kernel takes the codes it has every op for (its program is lowered from the plain one) from 48 leaves
        // on: its fixed cost per pass (four passes over the channel row, the epilogue) wants a tree to spread over.  Measured at the
        // end of round 3 (tools/lane_threshold.py, N = 1024, L = 8, 2^18 rows, lane against path-group kernels): 44 leaves (K = 64)
        // 59.5 against 63.7 M cw/s, 50 leaves (K = 100) 55.4 against 48.9, cfg 4 (75 leaves) 45.0 against 40.2, cfg 3 (95) 34.0
        // against 27.8.  (Until the leaves rode on the walks the line was 48 FORKING leaves and cfg 4 stayed on the path groups.)
        // (The int8 decoder SclFipChar runs on the same kernel -- saturating f / g, repetition leaves at every size, int8 rows
        //  widened to fp32 on the way in: scl_lane_char.hip --, cfg3_i8 27.4 against 23.4 M cw/s on the path groups; the same
        //  rule fits it, tools/lane_threshold.py char: 38 leaves 43.8 against 44.1, 53 leaves 38.1 against 32.8 M cw/s.)
        // Round 4: the same kernel under other geometries (scl_lane_geo.h) -- N = 512 / 2048 / 4096 and 16 / 32 lanes per codeword;
        // the default geometry keeps its rule, the others take every tree the lowering can express from 48 leaves on.
        const bool geo_default = d->logN == 10 && d->L > 4 && d->L <= 8;
        // N = 256 / 512 (tools/lane_threshold_small_n.py, 2^18 rows, lane against path groups, M cw/s): the fixed cost of a pass is smaller --
        // no or one level above stage 8 -- and the line sits lower: N = 256 L = 8: 16 leaves (K = 16) 237 / 284, 22 leaves (K = 32) 200 / 178,
        // 9 leaves at rate 0.96 169 / 144; N = 512 L = 8: 23 leaves (K = 24) 113 / 158, 32 leaves (K = 48) 110 / 104, 17 leaves at rate 0.94
        // 75.5 / 67.8; L = 2: the lane kernels win every one of those (N = 256 K = 16: 685 / 584).
        // N = 128: every tree (L = 8: K = 8 575 / 454, K = 64 272 / 217; L = 2: K = 64 1 096 / 705).
        // (N = 64 likewise: L = 8 K = 32 430 / 331, K = 8 668 / 533; L = 2 K = 32 1 693 / 1 078)
        // N = 1024 with 2 / 4 lanes: K = 64 (44 leaves) L = 2 144 / 111 -- from 40 leaves.
        const unsigned min_leaves = d->logN >= 10 ? ((d->logN == 10 && d->L <= 4) ? 40u : 48u)
                                                  : (d->logN <= 7 || d->L <= 4 || 2u * d->K >= d->N) ? 0u : (d->logN == 9 ? 28u : 20u);
        if (!small && !geo_default && pamd::find_lane_kernel(d->logN, d->L, chr, false) && !PAMD_ENV("PCD_NO_LANE") &&
            (d->scl_prog.leaves >= min_leaves || PAMD_ENV("PCD_FORCE_LANE"))) {
            bool big = false, hi = false;
            if (pamd::lower_lane(d->logN, d->scl_prog, d->lane_prog, chr, &big, &hi)) {
                const pamd::LaneKernel* lk = pamd::find_lane_kernel(d->logN, d->L, chr, hi);
                // (the leaves above stage 8 are recomputed from the channel row, which 32 / 16 codewords per wave do not pay at the low rates:
                //  N = 2048 L = 2 K = 128 38.6 / 47.0, K = 400 36.2 / 37.9, K = 512 35.8 / 32.2; N = 4096 L = 4 K = 256 13.6 / 15.2, K = 512
                //  11.8 / 12.6, K = 1024 10.1 / 8.8 M cw/s lane / path groups)
                //  (... and at N = 4096 at rate 1/32 with any number of lanes: K = 128 L = 8 10.0 / 11.6, L = 32 2.9 / 3.1)
                if (lk && hi && ((4u * d->K < d->N && (lk->lcl == 1 || (d->logN == 12 && lk->lcl == 2))) || (d->logN == 12 && 32u * d->K <= d->N)) &&
                    !PAMD_ENV("PCD_FORCE_LANE"))
                    lk = nullptr;
                if (lk) {
                    d->has_lane = true;
                    d->lane_k = lk;
                    d->lane_cpw = 64u >> lk->lcl;
                }
            }
        }
        if (!small && geo_default && (!chr || !PAMD_ENV("PCD_NO_CHAR_LANE")) && !PAMD_ENV("PCD_NO_LANE")) {
            // (... or from 30 FORKING leaves on: the high-rate trees have few leaves, but large RateOne / SPC ones -- K = 900: 46 leaves, 35 of
            //  them forking, 35.1 against 24.9 M cw/s)
            unsigned forks = 0;
            for (const auto& in : d->scl_prog.ins) forks += (in.op == pamd::SCL_R1 || in.op == pamd::SCL_SPC || in.op == pamd::SCL_REP) ? 1u : 0u;
            // (the second clause only from rate 1/2 on: K = 64 has 44 leaves, 43 of them forking, and runs 63.1 against 64.5 M cw/s)
            if (d->scl_prog.leaves >= 48u || (forks >= 30u && 2u * d->K >= d->N) || PAMD_ENV("PCD_FORCE_LANE"))
                d->has_lane = pamd::lower_lane(d->logN, d->scl_prog, d->lane_prog, chr, &d->lane_big);
        }
    }
    compute_geometry(d);
    if ((size_t)d->sc_wave_words * 4 * (64u >> d->sc_lcl) > kMaxLds && d->has_sc) {
        delete d;
        return fail(PCD_ERR_CAPACITY, "block length too large for the SC kernel's LDS layout");
    }
    if (d->has_scl) {
        const unsigned fb = ilog2(d->lmax) - d->cwl;
        if (d->logN * fb > 64) {
            const std::string msg = "(blockLength, listSize) does not fit the SCL kernel: log2(blockLength) * log2(listSize) = " +
                                    std::to_string(d->logN * fb) + " bits of path bookkeeping exceed the 64-bit slot word";
            delete d;
            return fail(PCD_ERR_CAPACITY, msg);
        }
        if ((size_t)d->scl_wave_words * 4 > kMaxLds) {
            const std::string msg = "(blockLength, listSize) does not fit the SCL kernel: needs " +
                                    std::to_string((size_t)d->scl_wave_words * 4) + " B of LDS per codeword (max 163840)";
            delete d;
            return fail(PCD_ERR_CAPACITY, msg);
        }
    }
    *out = d;
    return PCD_OK;
}

void pcd_destroy(pcd_decoder* d)
{
    if (!d) return;
    if (d->small) pcd_destroy(d->small);
    if (d->dev_ready) {
        pamd::DeviceScope scope;
        (void)scope.enter(d->device);
        (void)hipFree(d->d_sc_prog);
        (void)hipFree(d->d_scl_prog);
        if (d->d_lane_prog) (void)hipFree(d->d_lane_prog);
        if (d->d_pair_prog) (void)hipFree(d->d_pair_prog);
        if (d->d_ext) (void)hipFree(d->d_ext);
        if (d->d_syn_bytes) (void)hipFree(d->d_syn_bytes);
        (void)hipFree(d->d_lut);
        (void)hipFree(d->d_syn);
        (void)hipFree(d->d_count);
        for (auto e : d->ev)
            if (e) (void)hipEventDestroy(e);
        if (d->last_ev) (void)hipEventDestroy(d->last_ev);
    }
    if (d->d_in) (void)hipFree(d->d_in);
    if (d->d_i8) (void)hipFree(d->d_i8);
    for (int b = 0; b < pcd_decoder::kRing; ++b) {
        if (d->h_ring[b]) (void)hipHostFree(d->h_ring[b]);
        if (d->d_ring[b]) (void)hipFree(d->d_ring[b]);
        if (d->ev_h2d[b]) (void)hipEventDestroy(d->ev_h2d[b]);
        if (d->ev_run[b]) (void)hipEventDestroy(d->ev_run[b]);
    }
    if (d->s_in) (void)hipStreamDestroy(d->s_in);
    if (d->s_run) (void)hipStreamDestroy(d->s_run);
    if (d->s_out) (void)hipStreamDestroy(d->s_out);
    if (d->h_res) (void)hipHostFree(d->h_res);
    if (d->d_out) (void)hipFree(d->d_out);
    if (d->d_ok) (void)hipFree(d->d_ok);
    if (d->d_met) (void)hipFree(d->d_met);
    if (d->d_list) (void)hipFree(d->d_list);
    if (d->d_arena) (void)hipFree(d->d_arena);
    delete d;
}

int pcd_set_error_detection(pcd_decoder* d, unsigned size, const char* type)
{
    if (!d || !type) return fail(PCD_ERR_INVALID_ARG, "null argument");
    pamd::Detector det;
    std::string err;
    const int rc = pamd::make_detector(size, type, det, err);
    if (rc) return fail(rc, err);
    d->det = det;
    d->det_dirty = true;
    if (d->small) {
        d->small->det = det;
        d->small->det_dirty = true;
    }
    return PCD_OK;
}

int pcd_set_systematic(pcd_decoder* d, int systematic)
{
    if (!d) return fail(PCD_ERR_INVALID_ARG, "null decoder");
    d->systematic = systematic != 0;
    if (d->small) d->small->systematic = d->systematic;
    return PCD_OK;
}

int pcd_set_small_batch_rows(pcd_decoder* d, size_t rows)
{
    if (!d) return fail(PCD_ERR_INVALID_ARG, "null decoder");
    if (d->small) d->small_rows = rows;
    return PCD_OK;
}
size_t pcd_small_batch_rows(const pcd_decoder* d) { return (d && d->small) ? d->small_rows : 0; }

int pcd_is_systematic(const pcd_decoder* d) { return d && d->systematic; }
size_t pcd_block_length(const pcd_decoder* d) { return d ? d->N : 0; }
size_t pcd_info_length(const pcd_decoder* d) { return d ? d->K : 0; }
size_t pcd_list_size(const pcd_decoder* d) { return d ? d->L : 0; }
size_t pcd_out_bytes(const pcd_decoder* d) { return d ? d->nb : 0; }

size_t pcd_frozen_bits(const pcd_decoder* d, uint32_t* out, size_t cap)
{
    if (!d) return 0;
    if (out) std::memcpy(out, d->frozen.data(), std::min(cap, d->frozen.size()) * sizeof(uint32_t));
    return d->frozen.size();
}

int pcd_error_detection_mode(const pcd_decoder* d, char* buf, size_t cap)
{
    if (!d || !buf || !cap) return fail(PCD_ERR_INVALID_ARG, "null argument");
    std::snprintf(buf, cap, "%s-%u", d->det.type_name().c_str(), d->det.size);
    return PCD_OK;
}

int pcd_decode_f32(pcd_decoder* d, const float* llr, size_t rows, uint8_t* out, uint8_t* ok, float* metrics,
                   int mem, void* stream)
{
    return decode_any(d, llr, false, rows, out, ok, metrics, mem, stream);
}

int pcd_decode_i8(pcd_decoder* d, const int8_t* llr, size_t rows, uint8_t* out, uint8_t* ok, float* metrics,
                  int mem, void* stream)
{
    return decode_any(d, llr, true, rows, out, ok, metrics, mem, stream);
}

int pcd_program_stats(const pcd_decoder* d, int list, unsigned counts[2])
{
    if (!d || !counts) return fail(PCD_ERR_INVALID_ARG, "null argument");
    const pamd::Program* p = list ? (d->has_scl ? &d->scl_prog : nullptr) : (d->has_sc ? &d->sc_prog : nullptr);
    if (!p) return fail(PCD_ERR_INVALID_ARG, "this decoder has no such program");
    counts[0] = p->leaves;
    counts[1] = p->inner;
    return PCD_OK;
}

int pcd_launch_info(const pcd_decoder* d, int list, unsigned info[4])
{
    if (!d || !info) return fail(PCD_ERR_INVALID_ARG, "null argument");
    if (list) {
        if (!d->has_scl) return fail(PCD_ERR_INVALID_ARG, "this decoder has no SCL program");
        if (d->has_lane) {  // lane-per-path kernel: 40 KB of LDS per wave of 8 (4, 2) codewords, 1 lane per path, 4 waves per CU
            info[0] = 40960 / d->lane_cpw;
            info[1] = d->lane_cpw;
            info[2] = 64;  // path groups per wave: every lane is one
            info[3] = 4;
            return PCD_OK;
        }
        info[0] = (d->scl_wave_words * 4) >> d->cwl;
        info[1] = d->scl_wpb << d->cwl;
        info[2] = 64u >> d->lpl;
        info[3] = (unsigned)(kMaxLds / ((size_t)d->scl_wpb * d->scl_wave_words * 4));
    } else {
        if (!d->has_sc) return fail(PCD_ERR_INVALID_ARG, "this decoder has no SC program");
        if (d->has_pair) {  // two lanes per codeword: 40 KB of LDS per wave of 32 codewords, 4 waves per CU
            info[0] = 40960 / 32;
            info[1] = 32;
            info[2] = 1;
            info[3] = 4;
            return PCD_OK;
        }
        info[0] = d->sc_wave_words * 4;
        info[1] = d->sc_wpb * (64u >> d->sc_lcl);
        info[2] = 1;
        info[3] = (unsigned)(kMaxLds / ((size_t)info[1] * d->sc_wave_words * 4));
    }
    return PCD_OK;
}

int pcd_kernel_name(const pcd_decoder* d, int list, char* buf, size_t cap)
{
    if (!d || !buf || !cap) return fail(PCD_ERR_INVALID_ARG, "null argument");
    const char* name;
    if (list) {
        if (!d->has_scl) return fail(PCD_ERR_INVALID_ARG, "this decoder has no SCL program");
        if (d->has_lane)
            name = d->lane_k ? d->lane_k->name
                             : (d->dectype == 0 ? (d->lane_big ? "scl_lane_char_big_kernel" : "scl_lane_char_kernel")
                                                : (d->lane_big ? "scl_lane_big_kernel" : "scl_lane_kernel"));
        else
            name = "scl_decode_kernel";
    } else {
        if (!d->has_sc) return fail(PCD_ERR_INVALID_ARG, "this decoder has no SC program");
        name = d->has_pair ? (d->logN == 9 ? "sc_pair_n9_kernel" : d->logN == 8 ? "sc_pair_n8_kernel" : d->logN == 7 ? "sc_pair_n7_kernel" : d->logN == 6 ? "sc_pair_n6_kernel" : "sc_pair_kernel") : "sc_decode_kernel";
    }
    std::snprintf(buf, cap, "%s", name);
    return PCD_OK;
}

int pcd_set_timing(pcd_decoder* d, int enable)
{
    if (!d) return fail(PCD_ERR_INVALID_ARG, "null decoder");
    d->timing = enable != 0;
    d->ev_count = 0;
    return PCD_OK;
}

int pcd_kernel_times(pcd_decoder* d, float* ms, size_t cap, size_t* count)
{
    if (!d || !count) return fail(PCD_ERR_INVALID_ARG, "null argument");
    *count = d->ev_count;
    if (!ms) return PCD_OK;
    for (unsigned i = 0; i < d->ev_count && i < cap; ++i) {
        HIP_TRY(hipEventSynchronize(d->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms[i], d->ev[2 * i], d->ev[2 * i + 1]));
    }
    return PCD_OK;
}

#ifdef PAMD_DIAG
int pcd_scl_trace(pcd_decoder* d, const float* llr_row, float* trace, unsigned cap)
{
    if (!d || !llr_row || !trace) return fail(PCD_ERR_INVALID_ARG, "null argument");
    if (!d->has_scl) return fail(PCD_ERR_INVALID_ARG, "not a list decoder");
    pamd::DeviceScope scope;
    int rc = ensure_device(d, scope);
    if (rc) return rc;
    float* dtrace = nullptr;
    const size_t tb = (size_t)cap * (1 + d->L) * sizeof(float);
    if ((rc = grow(d->d_in, d->d_in_cap, (size_t)d->N * 4))) return rc;
    if ((rc = grow(d->d_out, d->d_out_cap, d->nb))) return rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dtrace), tb));
    HIP_TRY(hipMemset(dtrace, 0, tb));
    HIP_TRY(hipMemcpy(d->d_in, llr_row, (size_t)d->N * 4, hipMemcpyHostToDevice));
    pamd::DecodeParams P = base_params(d, d->d_in, 1, d->d_out, nullptr, nullptr);
    P.trace = dtrace;
    P.trace_cap = cap;
    rc = run_scl(d, P, false, nullptr);
    if (!rc) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(trace, dtrace, tb, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dtrace);
    return rc;
}
#endif  // PAMD_DIAG

#ifdef PAMD_DIAG
int pcd_scl_profile(pcd_decoder* d, const float* llr_row, uint8_t* ops, uint8_t* stages, uint32_t* cycles,
                    unsigned cap, unsigned* count)
{
    if (!d || !llr_row || !cycles || !count) return fail(PCD_ERR_INVALID_ARG, "null argument");
    if (!d->has_scl) return fail(PCD_ERR_INVALID_ARG, "not a list decoder");
    pamd::DeviceScope scope;
    int rc = ensure_device(d, scope);
    if (rc) return rc;
    const pamd::Program& prog = d->has_lane ? d->lane_prog : d->scl_prog;  // (the program the kernel of this decoder runs)
    const unsigned n = (unsigned)prog.ins.size();
    *count = n;
    uint32_t* dcyc = nullptr;
    size_t prows = 1;  // (PCD_PROFILE_ROWS = 8: llr_row holds 8 rows, one per codeword of the profiled wave)
    if (const char* e = PAMD_ENV("PCD_PROFILE_ROWS")) prows = (size_t)std::max(1, std::min(8, std::atoi(e)));
    if ((rc = grow(d->d_in, d->d_in_cap, prows * d->N * 4))) return rc;
    if ((rc = grow(d->d_out, d->d_out_cap, prows * d->nb))) return rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dcyc), (size_t)n * 5 * 4));
    HIP_TRY(hipMemset(dcyc, 0, (size_t)n * 5 * 4));
    HIP_TRY(hipMemcpy(d->d_in, llr_row, prows * d->N * 4, hipMemcpyHostToDevice));
    pamd::DecodeParams P = base_params(d, d->d_in, prows, d->d_out, nullptr, nullptr);
    P.cycles = dcyc;
    P.prog_len = n - 1;  // (the kernel's loop count and the offset of the phase stamps; the OP_END word is not executed)
    for (int rep = 0; rep < 2 && !rc; ++rep) rc = run_scl(d, P, false, nullptr);  // second run: warm caches
    if (!rc) {
        HIP_TRY(hipDeviceSynchronize());
        std::vector<uint32_t> h((size_t)n * 5);
        HIP_TRY(hipMemcpy(h.data(), dcyc, (size_t)n * 5 * 4, hipMemcpyDeviceToHost));
        if (const char* dump = PAMD_ENV("PCD_SUBPROFILE")) {  // leaf phase stamps (diagnostics)
            if (FILE* f = std::fopen(dump, "w")) {
                for (unsigned i = 0; i + 1 < n; ++i)
                    std::fprintf(f, "%u %u %u %u %u %u %u\n", (unsigned)prog.ins[i].op, (unsigned)prog.ins[i].stage,
                                 h[i], h[(n - 1) + 4 * i], h[(n - 1) + 4 * i + 1], h[(n - 1) + 4 * i + 2], h[(n - 1) + 4 * i + 3]);
                std::fclose(f);
            }
        }
        for (unsigned i = 0; i < n && i < cap; ++i) {
            cycles[i] = h[i];
            if (ops) ops[i] = prog.ins[i].op;
            if (stages) stages[i] = prog.ins[i].stage;
        }
    }
    (void)hipFree(dcyc);
    return rc;
}
#endif  // PAMD_DIAG

#ifdef PAMD_DIAG
int pcd_sc_profile(pcd_decoder* d, const float* llr_row, uint8_t* ops, uint8_t* stages, uint32_t* cycles,
                   unsigned cap, unsigned* count)
{
    if (!d || !llr_row || !cycles || !count) return fail(PCD_ERR_INVALID_ARG, "null argument");
    if (!d->has_sc) return fail(PCD_ERR_INVALID_ARG, "this decoder has no SC program");
    if (!d->has_pair && (d->sc_lcl != 4 || d->sc_arena_cw_words))
        return fail(PCD_ERR_UNSUPPORTED, "the SC cycle profile exists for the all-LDS kernel (N <= 1024) only");
    pamd::DeviceScope scope;
    int rc = ensure_device(d, scope);
    if (rc) return rc;
    const pamd::Program& prog = d->has_pair ? d->pair_prog : d->sc_prog;  // (the program the kernel of this decoder runs)
    const unsigned n = (unsigned)prog.ins.size();
    *count = n;
    uint32_t* dcyc = nullptr;
    if ((rc = grow(d->d_in, d->d_in_cap, (size_t)d->N * 4))) return rc;
    if ((rc = grow(d->d_out, d->d_out_cap, d->nb))) return rc;
    if ((rc = grow(d->d_ok, d->d_ok_cap, 4))) return rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dcyc), (size_t)n * 4));
    HIP_TRY(hipMemset(dcyc, 0, (size_t)n * 4));
    HIP_TRY(hipMemcpy(d->d_in, llr_row, (size_t)d->N * 4, hipMemcpyHostToDevice));
    pamd::DecodeParams P = base_params(d, d->d_in, 1, d->d_out, d->d_ok, nullptr);
    P.cycles = dcyc;
    for (int rep = 0; rep < 2 && !rc; ++rep) rc = run_sc(d, P, false, nullptr);  // second run: warm caches
    if (!rc) {
        HIP_TRY(hipDeviceSynchronize());
        std::vector<uint32_t> h(n);
        HIP_TRY(hipMemcpy(h.data(), dcyc, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < n && i < cap; ++i) {
            cycles[i] = h[i];
            if (ops) ops[i] = prog.ins[i].op;
            if (stages) stages[i] = prog.ins[i].stage;
        }
    }
    (void)hipFree(dcyc);
    return rc;
}
#endif  // PAMD_DIAG

const char* pcd_last_error(void) { return g_err.c_str(); }
const char* pcd_version(void) { return "polar_amd 0.1 (gfx950)"; }

}  // extern "C"
