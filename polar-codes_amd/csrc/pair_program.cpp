// Host-side lowering of the plain SC program to the program of the two-lanes-per-codeword kernel (sc_pair_kernel.hip):
// local stages (log2(block) - 1), the side of every finished block (right child or not) from its bit offset.
#include "program.h"

namespace pamd {

bool lower_pair(unsigned logN, const Program& sc, Program& out)
{
    out = Program();
    out.leaves = sc.leaves;
    out.inner = sc.inner;
    if (logN != 10) return false;
    auto put = [&](Op op, unsigned stage, unsigned side) {
        Ins i;
        i.op = (uint8_t)op;
        i.stage = (uint8_t)stage;
        i.side = (uint8_t)side;
        i.pad = 0;
        i.off = 0;
        out.ins.push_back(i);
    };
    for (const Ins& x : sc.ins) {
        if (x.op == OP_END) break;
        const unsigned s = x.stage;  // log2 of the block the op works on (the parent block for F / G / COMBINE ...)
        if (s < 2 || s > logN) return false;
        const unsigned T = s - 1;
        const unsigned side = s == logN ? 0u : ((x.off >> s) & 1u);  // the block [off, off + 2^s) is a right child
        auto leaf = [&](Op op, unsigned lo, unsigned hi) {
            if (T < lo || T > hi) return false;
            put(op, T, side);
            return true;
        };
        bool ok = true;
        switch (x.op) {
        case SC_F: if (T < 2) return false; put(PR_F, T, 0); break;
        case SC_G: if (T < 2) return false; put(PR_G, T, 0); break;
        case SC_G0R: if (T < 2) return false; put(PR_G0R, T, 0); break;
        case SC_COMBINE: if (T < 2) return false; put(PR_C, T, side); break;
        case SC_COMBINE0R: if (T < 2) return false; put(PR_C0R, T, side); break;
        case SC_RONE: if (T < 3 || T > 6) return false; put(PR_RONE, T, side); break;
        case SC_R0: ok = leaf(PR_R0, 1, 2); break;
        case SC_R1: ok = leaf(PR_R1, 1, 2); break;
        case SC_REP: ok = leaf(PR_REP, 1, 6); break;
        case SC_SPC: ok = leaf(PR_SPC, 1, 6); break;
        case SC_DREP: ok = leaf(PR_DREP, 1, 6); break;
        case SC_DSPC: ok = leaf(PR_DSPC, 3, 6); break;
        case SC_DSPC8: ok = leaf(PR_DSPC8, 2, 2); break;
        case SC_TREP: ok = leaf(PR_TREP, 3, 6); break;
        case SC_TYPE5: ok = leaf(PR_TYPE5, 2, 6); break;
        case SC_REPONE8: ok = leaf(PR_REPONE8, 2, 2); break;
        case SC_ZSPC8: ok = leaf(PR_ZSPC8, 2, 2); break;
        case SC_ZEROSPC: ok = leaf(PR_ZEROSPC, 3, 6); break;
        default: return false;
        }
        if (!ok) return false;
    }
    // the root must end in a COMBINE / COMBINE0R of the whole block: that is what fills the code word for the epilogue
    if (out.ins.empty() || (out.ins.back().op != PR_C && out.ins.back().op != PR_C0R) || out.ins.back().stage != logN - 1) return false;
    put(OP_END, 0, 0);
    put(OP_END, 0, 0);  // (the kernel fetches one instruction ahead)
    return true;
}

}  // namespace pamd
