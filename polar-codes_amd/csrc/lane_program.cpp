// Host-side lowering of the plain list-decoder program to the program of the lane-per-path kernel
// (scl_lane_kernel.hip): the root and its children are folded into LN_VT passes over the channel row, the G behind an
// all-frozen left child becomes LN_G0, the root's COMBINE is left to the kernel's epilogue.
#include "program.h"

namespace pamd {

bool lower_lane(unsigned logN, const Program& scl, Program& out, bool chr, bool* needs_big, bool* needs_hi)
{
    if (needs_hi) *needs_hi = false;
    out = Program();
    out.leaves = scl.leaves;
    out.inner = scl.inner;
    if (needs_big) *needs_big = false;
    if (logN < 6 || logN > 12) return false;
    const std::vector<Ins>& v = scl.ins;
    auto put = [&](Op op, unsigned stage, unsigned side, unsigned off) {
        Ins i;
        i.op = (uint8_t)op;
        i.stage = (uint8_t)stage;
        i.side = (uint8_t)side;
        i.pad = 0;
        i.off = off;
        out.ins.push_back(i);
    };
    bool right_half = false;  // the root's G has been seen
    bool isg_at[16] = { false };  // other block lengths: what the last op seen at stage t >= 9 was (true: a G)
    // N >= 2048: a leaf of 512 / 1024 LLRs -- the kernel recomputes its block from the channel row (off bit l = the op at stage
    // logN - l is a G, for the stages logN .. stage + 1)
    auto high_leaf = [&](const Ins& x) { return x.stage > 8 && logN >= 11 && x.stage <= 10 && x.stage < logN; };
    auto high_gm = [&](unsigned stage) {
        unsigned gm = 0;
        for (unsigned t = stage + 1u; t <= logN; ++t)
            if (isg_at[t]) gm |= 1u << (logN - t);
        return gm;
    };
    for (size_t k = 0; k < v.size(); ++k) {
        const Ins& x = v[k];
        switch (x.op) {
        case OP_END: break;
        case SCL_F:
        case SCL_G: {
            const bool isg = x.op == SCL_G;
            if (logN == 8 && x.stage == 8 && !isg) {  // N = 256: stage 8 is the channel row -- the first pass copies it into the arena column of the
                put(LN_VT, 8, x.side, 0);             // one live path and does this F; the root's G is an ordinary G of stage 8
                continue;
            }
            if (logN <= 7 && x.stage == logN && !isg) put(LN_VT, logN, x.side, 0);  // N = 128: stage 7 is the channel row -- copied into the LDS column of the one live path; the F follows as an ordinary F
            if (logN != 10 && x.stage >= 9) {
                // N = 512 / 2048 / 4096: every stage above 8 is virtual -- the op of stage 9 becomes ONE pass over the channel row
                // that recomputes the stages in between (op_vtg); off bit l = the op at stage logN - l is a G
                if (x.stage > logN) return false;
                isg_at[x.stage] = isg;
                if (x.stage > 9) {  // must be followed by the F of the stage below, or by an all-frozen leaf of 512 / 1024 LLRs
                    if (k + 1 >= v.size() || v[k + 1].stage + 1u != x.stage) return false;
                    const uint8_t nop = v[k + 1].op;
                    const bool leaf_high = (nop == SCL_R0 || nop == SCL_R1 || nop == SCL_SPC || (chr && nop == SCL_REP)) && logN >= 11 && v[k + 1].stage <= 10 &&
                                           v[k + 1].stage < logN;
                    if (nop != SCL_F && !leaf_high) return false;
                    continue;
                }
                unsigned gm = 0;
                for (unsigned t = 9; t <= logN; ++t)
                    if (isg_at[t]) gm |= 1u << (logN - t);
                if (k + 1 < v.size() && v[k + 1].op == SCL_F && v[k + 1].stage == 8) ++k;  // (the F of stage 8 is part of the pass)
                put(LN_VT, x.stage, x.side, gm);
                continue;
            }
            if (logN == 10 && x.stage == logN) {  // root: must be followed by the F of its child (stage 9 is virtual)
                if (isg) right_half = true;
                if (k + 1 >= v.size() || v[k + 1].op != SCL_F || v[k + 1].stage != logN - 1) return false;
                continue;  // folded into the LN_VT emitted for that F
            }
            if (logN == 10 && x.stage == logN - 1) {
                unsigned off = (right_half ? 1u : 0u) | (isg ? 2u : 0u);
                if (k + 1 < v.size() && v[k + 1].op == SCL_F && v[k + 1].stage == logN - 2) {
                    off |= 4u;
                    ++k;
                }
                put(LN_VT, x.stage, x.side, off);
                continue;
            }
            if (x.stage < 2) return false;
            // G behind an all-frozen left sibling: ... F(s) R0(s-1) G(s)
            if (isg && k >= 1 && v[k - 1].op == SCL_R0 && v[k - 1].stage + 1u == x.stage) {
                put(LN_G0, x.stage, x.side, 0);
                continue;
            }
            put(isg ? LN_G : LN_F, x.stage, x.side, 0);
            break;
        }
        case SCL_COMBINE:
            if (x.stage == logN) break;  // the kernel's epilogue assembles the code word
            if (x.stage < 2) return false;
            put(LN_C, x.stage, x.side, 0);
            break;
        case SCL_R0:
            if (logN <= 8 && x.stage >= logN) return false;  // (N = 256 / 128: a leaf that is the whole code)
            if (high_leaf(x)) {
                put(LN_R0, x.stage, x.side, high_gm(x.stage));
                if (needs_hi) *needs_hi = true;
                break;
            }
            if (x.stage > 8 || x.stage < 1) return false;
            put(LN_R0, x.stage, x.side, 0);
            break;
        case SCL_R1:
            if (logN <= 8 && x.stage >= logN) return false;
            if (high_leaf(x)) {
                put(LN_R1, x.stage, x.side, high_gm(x.stage));
                if (needs_hi) *needs_hi = true;
                break;
            }
            if (x.stage > 8 || x.stage < 1) return false;
            if (x.stage == 8 && needs_big) *needs_big = true;  // (a forking leaf of 256 LLRs: the BIG instantiation of the kernel)
            put(LN_R1, x.stage, x.side, 0);
            break;
        case SCL_SPC:
            if (logN <= 8 && x.stage >= logN) return false;
            if (high_leaf(x)) {
                put(LN_SPC, x.stage, x.side, high_gm(x.stage));
                if (needs_hi) *needs_hi = true;
                break;
            }
            if (x.stage > 8 || x.stage < 2) return false;
            if (x.stage == 8 && needs_big) *needs_big = true;
            put(LN_SPC, x.stage, x.side, 0);
            break;
        case SCL_REP:
            if (logN <= 8 && x.stage >= logN) return false;
            if (chr && high_leaf(x)) {
                put(LN_REP, x.stage, x.side, high_gm(x.stage));
                if (needs_hi) *needs_hi = true;
                break;
            }
            if (x.stage > (chr ? 8u : 2u) || x.stage < 1) return false;  // (int8 decoder: repetition leaves at every size)
            put(LN_REP, x.stage, x.side, 0);
            break;
        default: return false;
        }
    }
    put(OP_END, 0, 0, 0);
    // the first op must be the left-left pass over the channel row -- or, at N >= 2048, an all-frozen leaf above stage 8: then
    // the virtual rows that pass would have produced come from an op of their own
    if (out.ins.empty()) return false;
    if (out.ins[0].op != LN_VT) {
        if (!(logN >= 11 && out.ins[0].op >= LN_R0 && out.ins[0].op <= LN_SPC && out.ins[0].stage > 8)) return false;
        Ins i;
        i.op = (uint8_t)LN_VROWS;
        i.stage = (uint8_t)logN;
        i.side = 0;
        i.pad = 0;
        i.off = 0;
        out.ins.insert(out.ins.begin(), i);
    }
    // fuse what lies between two leaves into LN_WALK: [C ... C] [G | G0] [F ... F]
    std::vector<Ins> w;
    const std::vector<Ins>& o = out.ins;
    for (size_t k = 0; k < o.size();) {
        if (o[k].op != LN_C && o[k].op != LN_G && o[k].op != LN_G0 && o[k].op != LN_F) {
            w.push_back(o[k++]);
            continue;
        }
        unsigned c0 = 0, c1 = 0, gs = 0, gk = 0, f1 = 0, f0 = 0, side = 0;
        if (o[k].op == LN_C) {
            c0 = c1 = o[k].stage;
            side = o[k].side;
            ++k;
            while (o[k].op == LN_C && o[k].stage == c1 + 1 && side == 1) {  // (the node below was this one's right child)
                c1 = o[k].stage;
                side = o[k].side;
                ++k;
            }
        }
        if (o[k].op == LN_G || o[k].op == LN_G0) {
            gs = o[k].stage;
            gk = o[k].op == LN_G ? 1u : 2u;
            ++k;
        }
        if (o[k].op == LN_F && (gs == 0 || o[k].stage + 1u == gs)) {
            f1 = f0 = o[k].stage;
            ++k;
            while (o[k].op == LN_F && o[k].stage + 1u == f0) {
                f0 = o[k].stage;
                ++k;
            }
        }
        // the leaf the walk ends at rides along (one dispatch fewer; its trace record is the walk's)
        unsigned r0s = 0, r0side = 0, lkind = 0;
        if ((o[k].op == LN_R0 || o[k].op == LN_R1 || o[k].op == LN_SPC || o[k].op == LN_REP) && o[k].stage >= 1 &&
            ((f1 != 0 && o[k].stage + 1u == f0) || (f1 == 0 && gs != 0 && o[k].stage + 1u == gs))) {
            lkind = o[k].op == LN_R0 ? 1u : (o[k].op == LN_R1 ? 2u : (o[k].op == LN_SPC ? 3u : 4u));
            r0s = o[k].stage;
            r0side = o[k].side;
            ++k;
        }
        Ins i;
        i.op = (uint8_t)LN_WALK;
        i.stage = 0;
        i.side = (uint8_t)side;
        i.pad = 0;
        i.off = c0 | (c1 << 4) | (gs << 8) | (gk << 12) | (f1 << 16) | (f0 << 20) | (r0s << 24) | (r0side << 28) | (lkind << 29);
        w.push_back(i);
    }
    w.push_back(w.back());  // a spare word behind OP_END: the kernel fetches one instruction ahead
    out.ins.swap(w);
    return true;
}

}  // namespace pamd
