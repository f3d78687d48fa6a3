// Host-side lowering of the plain SC program to the program of the two-lanes-per-codeword kernel (sc_pair_kernel.hip):
// local stages (log2(block) - 1), the side of every finished block (right child or not) from its bit offset.
#include "program.h"

namespace pamd {

bool lower_pair(unsigned logN, const Program& sc, Program& out, bool chr, bool* needs_big)
{
    out = Program();
    out.leaves = sc.leaves;
    out.inner = sc.inner;
    bool big = false;  // a leaf above 128 positions: the BIG instantiations of the kernel
    if (needs_big) *needs_big = false;
    if (logN < 6 || logN > 10) return false;  // (N = 64 / 128 / 256 / 512: sc_pair_n6 / n7 / n8 / n9.hip)
    auto put = [&](Op op, unsigned stage, unsigned side) {
        Ins i;
        i.op = (uint8_t)op;
        i.stage = (uint8_t)stage;
        i.side = (uint8_t)side;
        i.pad = 0;
        i.off = 0;
        out.ins.push_back(i);
    };
    for (size_t xi = 0; xi < sc.ins.size(); ++xi) {
        const Ins& x = sc.ins[xi];
        if (x.op == OP_END) break;
        const unsigned s = x.stage;  // log2 of the block the op works on (the parent block for F / G / COMBINE ...)
        if (s < 2 || s > logN) return false;
        const unsigned T = s - 1;
        const unsigned side = s == logN ? 0u : ((x.off >> s) & 1u);  // the block [off, off + 2^s) is a right child
        auto leaf = [&](Op op, unsigned lo, unsigned hi) {
            if (T < lo || T > hi) return false;
            if (T >= 7) big = true;
            put(op, T, side);
            return true;
        };
        bool ok = true;
        switch (x.op) {
        case SC_F: if (T < 2) return false; put(PR_F, T, 0); break;
        case SC_G: if (T < 2) return false; put(PR_G, T, 0); break;
        case SC_G0R:
            if (chr && s == 2) {  // int8 decoder: a block of 4 whose left half is frozen and whose right half is free, as ONE leaf
                if (xi + 2 >= sc.ins.size() || sc.ins[xi + 1].op != SC_R1 || sc.ins[xi + 1].stage != 1 || sc.ins[xi + 2].op != SC_COMBINE0R ||
                    sc.ins[xi + 2].stage != 2)
                    return false;
                put(PR_ZR1, 1, side);
                xi += 2;
                break;
            }
            if (T < 2) return false;
            put(PR_G0R, T, 0);
            break;
        case SC_COMBINE: if (T < 2) return false; put(PR_C, T, side); break;
        case SC_COMBINE0R: if (T < 2) return false; put(PR_C0R, T, side); break;
        case SC_RONE: if (T < 2) return false; put(PR_RONE, T, side); break;
        case SC_R0: ok = leaf(PR_R0, 1, 2); break;
        case SC_R1: ok = leaf(PR_R1, 1, 2); break;
        case SC_REP: ok = leaf(PR_REP, 1, 8); break;
        case SC_SPC: ok = leaf(PR_SPC, 1, 8); break;
        case SC_DREP: ok = leaf(PR_DREP, 1, 8); break;
        case SC_DSPC: ok = leaf(PR_DSPC, 3, 6); break;
        case SC_DSPC8: ok = leaf(PR_DSPC8, 2, 2); break;
        case SC_TREP: ok = leaf(PR_TREP, 3, 8); break;
        case SC_TYPE5: ok = leaf(PR_TYPE5, 2, 8); break;
        case SC_REPONE8: ok = leaf(PR_REPONE8, 2, 2); break;
        case SC_ZSPC8: ok = leaf(PR_ZSPC8, 2, 2); break;
        case SC_ZEROSPC: ok = leaf(PR_ZEROSPC, 3, 6); break;
        case SCC_REP: ok = chr && leaf(PR_REP, 1, 8); break;
        case SCC_SPC: ok = chr && leaf(PR_SPC, 1, 8); break;
        case SCC_DREP: ok = chr && leaf(PR_DREP, 4, 8); break;
        case SCC_ZEROSPC: ok = chr && leaf(PR_ZEROSPC, 2, 6); break;
        default: return false;
        }
        if (!ok) return false;
    }
    // the root must end in a COMBINE / COMBINE0R of the whole block: that is what fills the code word for the epilogue
    if (out.ins.empty() || (out.ins.back().op != PR_C && out.ins.back().op != PR_C0R && out.ins.back().op != PR_RONE) || out.ins.back().stage != logN - 1)
        return false;
    // Fuse everything between two leaves into one PR_WALK:
    //   word 0: op | leaf kind (op - PR_R0 + 1, 0 = none) << 8 | c0 << 12 | steps up << 16 | d1 << 20 | steps down << 24 | first step down << 28
    //           (0 F, 1 G, 2 G0R)
    //   word 1: 2 bits per step up (1 COMBINE, 2 COMBINE0R, 3 RONE; stages c0, c0 + 1, ...) | bit 16 + i: step i > 0 down is a G0R
    //           (stages d1, d1 - 1, ...)
    // The leaf's stage (one below the last step down) and every side (a finished block is a right child unless a RONE, the G or
    // the end follows; the leaf is a left child behind an F) follow from the steps.
    std::vector<Ins> plain;
    plain.swap(out.ins);
    size_t k = 0;
    const size_t n = plain.size();
    auto is_up = [](uint8_t op) { return op == PR_C || op == PR_C0R || op == PR_RONE; };
    while (k < n) {
        uint32_t c0 = 0, nup = 0, upk = 0, d1 = 0, ndn = 0, first = 0, dnk = 0, lk = 0;
        while (k < n && is_up(plain[k].op)) {
            if (nup == 0) c0 = plain[k].stage;
            else if (plain[k].stage != c0 + nup) return false;
            if (nup >= 8) return false;
            upk |= (plain[k].op == PR_C ? 1u : plain[k].op == PR_C0R ? 2u : 3u) << (2 * nup);
            ++nup;
            ++k;
        }
        if (k < n) {
            if (plain[k].op != PR_F && plain[k].op != PR_G && plain[k].op != PR_G0R) return false;
            while (k < n && (plain[k].op == PR_F || plain[k].op == PR_G0R || (plain[k].op == PR_G && ndn == 0))) {
                if (ndn == 0) {
                    d1 = plain[k].stage;
                    first = plain[k].op == PR_F ? 0u : plain[k].op == PR_G ? 1u : 2u;
                } else {
                    if (plain[k].stage + ndn != d1) return false;
                    if (plain[k].op == PR_G0R) dnk |= 1u << ndn;
                }
                if (ndn >= 9) return false;
                ++ndn;
                ++k;
            }
            if (k >= n || plain[k].op < PR_R0 || plain[k].op > PR_ZR1) return false;
            if (plain[k].stage + ndn != d1) return false;
            lk = (uint32_t)plain[k].op - PR_R0 + 1u;
            ++k;
        }
        Ins w;
        w.op = PR_WALK;
        w.stage = (uint8_t)(lk | (c0 << 4));
        w.side = (uint8_t)(nup | (d1 << 4));
        w.pad = (uint8_t)(ndn | (first << 4));
        w.off = upk | (dnk << 16);
        out.ins.push_back(w);
    }
    put(OP_END, 0, 0);
    put(OP_END, 0, 0);  // (the kernel fetches one instruction ahead)
    if (needs_big) *needs_big = big;
    return true;
}

}  // namespace pamd
