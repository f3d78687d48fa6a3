// Host-only driver of the decode-program compilers (program.cpp, lane_program.cpp, pair_program.cpp): no HIP, no GPU.
// Built by `make -C polar-codes_amd hostcheck` with -fsanitize=address,undefined and run from tests/test_host.py:
//
//   hostcheck sets FILE     every frozen set of FILE ("N n f0 f1 ..." per line; '#' lines skipped): all compilers, float and int8
//   hostcheck fuzz COUNT SEED   COUNT random frozen sets (N = 8 .. 8192, every density, runs and scattered positions)
//   hostcheck dump N L chr < "f0 f1 ..."   the plain list program and its lane lowering, one op per line
//
// What "pass" means: no sanitizer report, every program ends in OP_END, stages and sides are in range, the lowered programs
// only use their own op codes, leaves + inner counts agree between the plain and the lowered program.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../csrc/program.h"

using namespace pamd;

static unsigned ilog2(unsigned n)
{
    unsigned s = 0;
    while ((1u << s) < n) ++s;
    return s;
}

static const char* opname(unsigned op)
{
    switch (op) {
    case OP_END: return "END";
    case SCL_F: return "F";
    case SCL_G: return "G";
    case SCL_COMBINE: return "C";
    case SCL_R0: return "R0";
    case SCL_R1: return "R1";
    case SCL_REP: return "REP";
    case SCL_SPC: return "SPC";
    case LN_VT: return "LN_VT";
    case LN_F: return "LN_F";
    case LN_G: return "LN_G";
    case LN_G0: return "LN_G0";
    case LN_C: return "LN_C";
    case LN_R0: return "LN_R0";
    case LN_R1: return "LN_R1";
    case LN_REP: return "LN_REP";
    case LN_SPC: return "LN_SPC";
    case LN_WALK: return "LN_WALK";
    case LN_VROWS: return "LN_VROWS";
    default: return "?";
    }
}

static int fail(const char* what, unsigned N, size_t nf)
{
    std::fprintf(stderr, "hostcheck: %s (N = %u, %zu frozen)\n", what, N, nf);
    return 1;
}

static int check_program(const Program& p, unsigned logN, unsigned lo, unsigned hi, const char* what, unsigned N, size_t nf)
{
    if (p.ins.empty()) return fail(what, N, nf);
    bool ended = false;
    for (const Ins& i : p.ins) {
        if (i.op == OP_END) {
            ended = true;
            continue;
        }
        if (i.op < lo || i.op > hi) return fail(what, N, nf);
        if (i.op != LN_WALK && i.op != PR_WALK && i.stage > logN) return fail(what, N, nf);
    }
    return ended ? 0 : fail(what, N, nf);
}

static int run_set(unsigned N, const std::vector<uint32_t>& frozen, unsigned long* stats)
{
    const unsigned logN = ilog2(N);
    for (int chr = 0; chr < 2; ++chr) {
        Program sc, scl, lane, pair;
        const std::string e = compile_sc(N, frozen, sc, chr != 0);
        if (e.empty()) {
            ++stats[0];
            if (check_program(sc, logN, SC_F, SCC_ZEROSPC, "SC program malformed", N, frozen.size())) return 1;
            bool big = false;
            if (lower_pair(logN, sc, pair, chr != 0, &big)) {
                ++stats[1];
                if (check_program(pair, logN, PR_F, PR_WALK, "pair program malformed", N, frozen.size())) return 1;
                if (pair.leaves != sc.leaves || pair.inner != sc.inner) return fail("pair program counts differ", N, frozen.size());
            }
        } else {
            ++stats[2];
        }
        compile_scl(N, frozen, scl, chr != 0);
        ++stats[3];
        if (check_program(scl, logN, SCL_F, SCL_SPC, "SCL program malformed", N, frozen.size())) return 1;
        bool big = false;
        if (lower_lane(logN, scl, lane, chr != 0, &big)) {
            ++stats[4];
            if (check_program(lane, logN, LN_VT, LN_VROWS, "lane program malformed", N, frozen.size())) return 1;
            if (lane.leaves != scl.leaves || lane.inner != scl.inner) return fail("lane program counts differ", N, frozen.size());
            // every leaf of the plain program must be reachable in the lowered one: count them (riding on walks or on their own)
            unsigned leaves = 0;
            for (const Ins& i : lane.ins) {
                if (i.op >= LN_R0 && i.op <= LN_SPC) ++leaves;
                if (i.op == LN_WALK && ((i.off >> 29) & 7u)) ++leaves;
                if (i.op == OP_END) break;
            }
            if (leaves != scl.leaves) return fail("lane program lost a leaf", N, frozen.size());
        }
    }
    // the tables the kernels' epilogues read
    const std::vector<DepositWord> dep = deposit_table(N, frozen);
    if (dep.size() != (N + 31) / 32) return fail("deposit table size", N, frozen.size());
    unsigned k = 0;
    for (const DepositWord& w : dep) {
        if (w.base != k) return fail("deposit table base", N, frozen.size());
        k += (unsigned)__builtin_popcount(w.mask);
        const uint32_t x = w.mask ? (__builtin_popcount(w.mask) == 32 ? 0xFFFFFFFFu : ((1u << __builtin_popcount(w.mask)) - 1u)) : 0u;
        if (deposit_apply(w, x) != w.mask) return fail("deposit table masks", N, frozen.size());
    }
    if (k != N - frozen.size()) return fail("deposit table count", N, frozen.size());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: hostcheck sets FILE | fuzz COUNT SEED | dump N L chr\n");
        return 2;
    }
    const std::string mode = argv[1];
    unsigned long stats[5] = { 0, 0, 0, 0, 0 };
    unsigned long sets = 0;
    if (mode == "sets" && argc >= 3) {
        std::ifstream f(argv[2]);
        if (!f) return fail("cannot open the file", 0, 0);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream is(line);
            unsigned N = 0;
            size_t n = 0;
            if (!(is >> N >> n)) continue;
            std::vector<uint32_t> fr(n);
            for (size_t i = 0; i < n; ++i) is >> fr[i];
            if (!is && n) return fail("short line", N, n);
            if (run_set(N, fr, stats)) return 1;
            ++sets;
        }
    } else if (mode == "fuzz" && argc >= 4) {
        const unsigned long count = std::strtoul(argv[2], nullptr, 10);
        std::mt19937_64 rng(std::strtoull(argv[3], nullptr, 10));
        for (unsigned long it = 0; it < count; ++it) {
            const unsigned logN = 3 + (unsigned)(rng() % 11);  // N = 8 .. 8192
            const unsigned N = 1u << logN;
            std::vector<uint32_t> fr;
            const unsigned kind = (unsigned)(rng() % 4);
            const double dens = (double)(rng() % 1001) / 1000.0;
            if (kind == 0) {  // scattered
                for (unsigned i = 0; i < N; ++i)
                    if ((double)(rng() % 100000) / 100000.0 < dens) fr.push_back(i);
            } else if (kind == 1) {  // polar-like: the density falls with the index weight
                for (unsigned i = 0; i < N; ++i) {
                    const double w = (double)__builtin_popcount(i) / (double)logN;
                    if ((double)(rng() % 100000) / 100000.0 < dens * (1.3 - w) + (w < 0.3 ? 0.5 : 0.0)) fr.push_back(i);
                }
            } else if (kind == 2) {  // runs: whole aligned blocks frozen or free (large leaves at every stage)
                unsigned i = 0;
                while (i < N) {
                    const unsigned lg = (unsigned)(rng() % (logN + 1));
                    const unsigned len = 1u << lg;
                    const unsigned start = i & ~(len - 1);
                    const unsigned end = std::min(N, start + len);
                    const bool frz = (rng() & 1u) != 0;
                    for (; i < end; ++i)
                        if (frz) fr.push_back(i);
                }
            } else {  // a prefix frozen, a few holes
                const unsigned cut = (unsigned)(dens * N);
                for (unsigned i = 0; i < N; ++i)
                    if (i < cut ? (rng() % 64 != 0) : (rng() % 64 == 0)) fr.push_back(i);
            }
            if (run_set(N, fr, stats)) return 1;
            ++sets;
        }
    } else if (mode == "dump" && argc >= 5) {
        const unsigned N = (unsigned)std::atoi(argv[2]);
        const bool chr = std::atoi(argv[4]) != 0;
        std::vector<uint32_t> fr;
        unsigned v;
        while (std::cin >> v) fr.push_back(v);
        Program scl, lane;
        compile_scl(N, fr, scl, chr);
        std::printf("# plain list program: %zu ops, %u leaves, %u inner\n", scl.ins.size(), scl.leaves, scl.inner);
        for (const Ins& i : scl.ins) std::printf("%s %u %u\n", opname(i.op), (unsigned)i.stage, (unsigned)i.side);
        bool big = false;
        const bool ok = lower_lane(ilog2(N), scl, lane, chr, &big);
        std::printf("# lane program: %s, %zu ops, big = %d\n", ok ? "ok" : "NOT EXPRESSIBLE", lane.ins.size(), (int)big);
        if (ok)
            for (const Ins& i : lane.ins) std::printf("%s %u %u 0x%08x\n", opname(i.op), (unsigned)i.stage, (unsigned)i.side, i.off);
        return 0;
    } else {
        std::fprintf(stderr, "usage: hostcheck sets FILE | fuzz COUNT SEED | dump N L chr\n");
        return 2;
    }
    std::printf("hostcheck ok: %lu sets; SC accepted %lu (pair kernel %lu), refused %lu; list programs %lu (lane kernel %lu)\n", sets, stats[0], stats[1],
                stats[2], stats[3], stats[4]);
    return 0;
}
