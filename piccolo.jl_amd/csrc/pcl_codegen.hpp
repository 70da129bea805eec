// pcl_codegen.hpp -- host-only: source generator for the PATTERN-COMPILED kernels.
//
// The generators of the reference's systems are sparse (multilevel transmons, config 3: 614 of 2916 entries) and exact
// iso(.) images  T = [[A, -B], [B, A]].  Every product of the Hessian of the Lagrangian acts on the state columns from the
// left, so one state column never needs another one: a lane can own one column, keep it in REGISTERS and apply G(u)^T as a
// straight line of fused multiply-adds whose register indices are the sparsity pattern and whose coefficients are scalar
// registers (s_load from a per-interval value table).  No LDS operand traffic, no matrix-core padding (a 54 x 54 x 27
// product costs 112 MFMAs = 7168 cycles of one SIMD; 307 FMAs = 1228 cycles).  The iso structure halves the code and the
// registers: lane (half, c) holds only ITS half x of column c (half 0: top rows a, half 1: bottom rows b), both halves run
// the SAME instructions  U = A^T x, V = B^T x  and one cross-half swap (v_permlane32_swap, no LDS) completes
//     top = A^T a + B^T b = U(half 0) + V(half 1),      bottom = -B^T a + A^T b = U(half 1) - V(half 0).
//
// The pattern is data, so the kernel source is generated per system and compiled on first use (hiprtc).
#pragma once

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace pcl_codegen {

struct SpPlan {
    int d = 0, m = 0, n = 0;
    int nz = 0, nzp = 0;          // union entries of the left column block [A; B] of G(u); padded: nz + kChunk rounded up to a multiple of 16
    std::vector<int> row, col;    // entry k (in the emission order of sp_gt): G[row][col], row < n, col < d
    std::vector<int> pos;         // row + n*col (column-major position in G0 / G_l)
    std::vector<double> coef;     // [nz][m]: G_l at pos
    std::vector<int> row_n, col_n, pos_n;  // the same entries in the emission order of sp_g (G(u) x: outputs by row)
    std::vector<double> coef_n;
    std::vector<int> doff;        // [m+1]: offsets of the drives' entries
    std::vector<int> drow, dcol;  // entries of the left column block of every G_l
    std::vector<int> dmagi;       // ... index of |value| in `mags`
    std::vector<char> dneg;       // ... value < 0
    std::vector<double> mags;     // distinct magnitudes of the drives' entries (the device table; they live in SGPRs for the whole launch)
    bool ok = false;
};

constexpr int kMaxMags = 8;  // distinct magnitudes of the drives' entries the kernels keep in scalar registers
constexpr int kChunk = 12;  // coefficients per asm statement of a table-driven product (8 + 4: s_load_dwordx16 + s_load_dwordx8)
constexpr int kGroup = 9;  // outputs accumulated together by the long product (independent dependency chains per wave; 3 .. 9 measure
                           // the same, 14 and more spill)

namespace detail {
struct Term {
    int out, in;   // output index, input index (< d: a[in], >= d: b[in - d])
    int src;       // caller's entry index; after ordering: the coefficient's table index
    bool neg;
    int mag = -1;  // >= 0: the coefficient is the resident magnitude mg.m<mag> instead of a table entry
};
// Emission order: the outputs in groups of `group`, round-robin inside a group (consecutive instructions belong to different
// dependency chains; a finished group is handed to the caller, so at most `group` accumulators are live).
static inline std::vector<Term> emission_order(const std::vector<Term> &terms, int d, int group) {
    std::vector<std::vector<Term>> by(d);
    for (const Term &t : terms) by[t.out].push_back(t);
    std::vector<Term> out;
    for (int g0 = 0; g0 < d; g0 += group) {
        size_t longest = 0;
        for (int o = g0; o < std::min(d, g0 + group); ++o) longest = std::max(longest, by[o].size());
        for (size_t t = 0; t < longest; ++t)
            for (int o = g0; o < std::min(d, g0 + group); ++o)
                if (t < by[o].size()) out.push_back(by[o][t]);
    }
    return out;
}
// top rows of T^T x = A^T a + B^T b:  o[c] += T[rho][c] * (rho < d ? a[rho] : b[rho - d])
static inline Term term_t(int row, int col, int d, int src) { return {col, row, src, false}; }
// T x on this half's x:  A block (rho < d): U[rho] += T[rho][c] x[c];  B block: V[rho - d] += T[rho][c] x[c]
// (top = U(0) - V(1), bottom = U(1) + V(0): the caller completes with -sgn)
static inline Term term_n(int row, int col, int d, int src) { return {row < d ? row : row - d, row < d ? col : d + col, src, false}; }

// Coefficients of a table-driven product travel in CHUNKS of (at most) eight consecutive table entries = one s_load_dwordx16.
// One volatile asm statement per chunk holds: the load of the NEXT chunk, this chunk's multiply-adds, the wait for that load.
// Everything that is in flight stays inside one statement: outside of them every register holds what the compiler thinks it
// holds.  (Issuing a load in one statement and waiting for it in a later one let the compiler re-use or spill the destination
// registers in between -- a scalar load then landed in a store's base address: memory fault at d = 31.  Left entirely to the
// compiler, the loads of a whole product are hoisted and hundreds of SGPRs are spilled to VGPR lanes.)
//
// Body of  template <class F> void f(x, tab, sgn, half, out):  `ordered` is in emission order with consecutive table indices
// starting at `base` (or resident magnitudes).  Entries of the A block (in < d) accumulate U[out] += coef x[in], entries of
// the B block V[out] += coef x[in - d]; a finished group of outputs is completed with the other half's V and passed to
// out(index, value).
// lds_exchange: the results go to an LDS tile anyway (Tl = the lane's slots, To = the other half's): U is written, the other
// half's slot receives -sgn V by an LDS atomic add (a wave's LDS operations complete in order), no cross-lane VALU work.
static inline void emit_groups(std::string &s, const std::vector<Term> &ordered, int d, int group, const char *tab, int base, bool lds_exchange = false) {
    char buf[512];
    const bool table = !ordered.empty() && ordered[0].mag < 0;
    // chunks: runs of <= 12 terms that do not straddle a group
    struct Run { size_t lo, hi; };
    std::vector<Run> runs;
    {
        size_t i = 0;
        for (int g0 = 0; g0 < d; g0 += group) {
            const int g1 = std::min(d, g0 + group);
            size_t j = i;
            while (j < ordered.size() && ordered[j].out < g1 && ordered[j].out >= g0) ++j;
            for (size_t r = i; r < j; r += kChunk) runs.push_back({r, std::min(j, r + kChunk)});
            i = j;
        }
    }
    // a chunk = 8 + 4 coefficients in two SGPR tuples pinned to physical registers (even chunks: s[36:51], s[52:59]; odd chunks:
    // s[60:75], s[76:83]) so that the asm text can name a coefficient's register pair and the tuple costs ONE operand
    auto regs16 = [](size_t c) { return c & 1 ? "s[60:75]" : "s[36:51]"; };
    auto regs8 = [](size_t c) { return c & 1 ? "s[76:83]" : "s[52:59]"; };
    auto coefreg = [](size_t c, int e) {
        static char b[24];
        const int base = (c & 1 ? 60 : 36) + (e < 8 ? 2 * e : 16 + 2 * (e - 8));
        snprintf(b, sizeof b, "s[%d:%d]", base, base + 1);
        return std::string(b);
    };
    if (table) {
        for (size_t c = 0; c < runs.size(); ++c) {
            snprintf(buf, sizeof buf, "%ssp_v8d k%zu; sp_v4d h%zu", c ? "; " : "    ", c, c);
            s += buf;
        }
        s += ";\n";
        if (!runs.empty()) {
            snprintf(buf, sizeof buf, "    asm volatile(\"s_load_dwordx16 %%0, %%2, %d\\n\\ts_load_dwordx8 %%1, %%2, %d\\n\\ts_waitcnt lgkmcnt(0)\" : \"=&{%s}\"(k0), \"=&{%s}\"(h0) : \"s\"(%s) : \"memory\");\n",
                     (base + (int)runs[0].lo) * 8, (base + (int)runs[0].lo + 8) * 8, regs16(0), regs8(0), tab);
            s += buf;
        }
    }
    size_t i = 0, run = 0;
    for (int g0 = 0; g0 < d; g0 += group) {
        const int g1 = std::min(d, g0 + group);
        s += "    {\n";
        std::vector<char> seen(2 * d, 0);
        for (int o = g0; o < g1; ++o) {
            snprintf(buf, sizeof buf, "        double u%d = 0.0, v%d = 0.0;\n", o, o);
            s += buf;
        }
        if (!table) {
            // resident coefficients: one volatile asm statement per multiply-add (the instruction order is the emission order; left
            // to schedule, the compiler floats the arithmetic away and spills)
            for (; i < ordered.size() && ordered[i].out < g1 && ordered[i].out >= g0; ++i) {
                const Term &q = ordered[i];
                const bool isv = q.in >= d;
                const int idx = isv ? q.in - d : q.in;
                char acc[16], mg[24];
                snprintf(acc, sizeof acc, "%c%d", isv ? 'v' : 'u', q.out);
                snprintf(mg, sizeof mg, "mg.m%d", q.mag);
                char &sn = seen[(isv ? d : 0) + q.out];
                if (!sn)
                    snprintf(buf, sizeof buf, "        asm volatile(\"v_mul_f64 %%0, %s%%1, %%2\" : \"=v\"(%s) : \"s\"(%s), \"v\"(x[%d]));\n", q.neg ? "-" : "", acc, mg, idx);
                else if (!q.neg)
                    snprintf(buf, sizeof buf, "        asm volatile(\"v_fmac_f64 %%0, %%1, %%2\" : \"+v\"(%s) : \"s\"(%s), \"v\"(x[%d]));\n", acc, mg, idx);
                else
                    snprintf(buf, sizeof buf, "        asm volatile(\"v_fma_f64 %%0, -%%1, %%2, %%0\" : \"+v\"(%s) : \"s\"(%s), \"v\"(x[%d]));\n", acc, mg, idx);
                sn = 1;
                s += buf;
            }
        } else {
            for (; run < runs.size() && runs[run].lo < ordered.size() && ordered[runs[run].lo].out < g1 && ordered[runs[run].lo].out >= g0; ++run) {
                const Run R = runs[run];
                const bool has_next = run + 1 < runs.size();
                // operands: [next chunk] accumulators ... | table pointer, coefficients, x values
                std::vector<std::string> outs, ins;
                std::vector<std::string> accn, xn;   // names already bound to an operand
                std::vector<int> acci, xi;
                std::string body;
                int nop = 0;
                if (has_next) {
                    snprintf(buf, sizeof buf, "\"=&{%s}\"(k%zu)", regs16(run + 1), run + 1);
                    outs.push_back(buf);
                    snprintf(buf, sizeof buf, "\"=&{%s}\"(h%zu)", regs8(run + 1), run + 1);
                    outs.push_back(buf);
                    nop += 2;
                }
                // first pass: bind accumulators (outputs come first in the operand numbering)
                struct Op { std::string acc; bool first; };
                std::vector<Op> ops;
                for (size_t t = R.lo; t < R.hi; ++t) {
                    const Term &q = ordered[t];
                    const bool isv = q.in >= d;
                    snprintf(buf, sizeof buf, "%c%d", isv ? 'v' : 'u', q.out);
                    const std::string acc = buf;
                    char &sn = seen[(isv ? d : 0) + q.out];
                    bool bound = false;
                    for (auto &a : accn) bound |= a == acc;
                    if (!bound) {
                        accn.push_back(acc);
                        acci.push_back(nop++);
                        outs.push_back(std::string(sn ? "\"+v\"(" : "\"=&v\"(") + acc + ")");
                    }
                    ops.push_back({acc, !sn});
                    sn = 1;
                }
                const int tab_op = nop++;
                snprintf(buf, sizeof buf, "\"s\"(%s)", tab);
                ins.push_back(buf);
                snprintf(buf, sizeof buf, "\"{%s}\"(k%zu)", regs16(run), run);  // this chunk's tuples: one operand each, named in the text
                ins.push_back(buf);
                snprintf(buf, sizeof buf, "\"{%s}\"(h%zu)", regs8(run), run);
                ins.push_back(buf);
                nop += 2;
                if (has_next) {
                    snprintf(buf, sizeof buf, "s_load_dwordx16 %%0, %%%d, %d\\n\\ts_load_dwordx8 %%1, %%%d, %d\\n\\t", tab_op, (base + (int)runs[run + 1].lo) * 8, tab_op,
                             (base + (int)runs[run + 1].lo + 8) * 8);
                    body += buf;
                }
                for (size_t t = R.lo; t < R.hi; ++t) {
                    const Term &q = ordered[t];
                    const bool isv = q.in >= d;
                    const int idx = isv ? q.in - d : q.in;
                    const std::string creg = coefreg(run, (int)(t - R.lo));
                    snprintf(buf, sizeof buf, "x[%d]", idx);
                    const std::string xname = buf;
                    int xop = -1;
                    for (size_t k = 0; k < xn.size(); ++k)
                        if (xn[k] == xname) xop = xi[k];
                    if (xop < 0) {
                        xop = nop++;
                        xn.push_back(xname);
                        xi.push_back(xop);
                        ins.push_back("\"v\"(" + xname + ")");
                    }
                    int aop = -1;
                    for (size_t k = 0; k < accn.size(); ++k)
                        if (accn[k] == ops[t - R.lo].acc) aop = acci[k];
                    if (ops[t - R.lo].first)
                        snprintf(buf, sizeof buf, "v_mul_f64 %%%d, %s%s, %%%d\\n\\t", aop, q.neg ? "-" : "", creg.c_str(), xop);
                    else if (!q.neg)
                        snprintf(buf, sizeof buf, "v_fmac_f64 %%%d, %s, %%%d\\n\\t", aop, creg.c_str(), xop);
                    else
                        snprintf(buf, sizeof buf, "v_fma_f64 %%%d, -%s, %%%d, %%%d\\n\\t", aop, creg.c_str(), xop, aop);
                    body += buf;
                }
                body += has_next ? "s_waitcnt lgkmcnt(0)" : "s_nop 0";
                s += "        asm volatile(\"" + body + "\" : ";
                for (size_t k = 0; k < outs.size(); ++k) s += (k ? ", " : "") + outs[k];
                s += " : ";
                for (size_t k = 0; k < ins.size(); ++k) s += (k ? ", " : "") + ins[k];
                s += " : \"memory\");\n";
                i = R.hi;
            }
        }
        if (lds_exchange) {
            // every U of the group is stored before any V is added: the slot a lane adds to is ANOTHER lane's store target, which
            // the compiler cannot see (per thread the two pointers never alias) -- the empty asm pins the order, the LDS pipeline
            // executes a wave's operations in it
            for (int o = g0; o < g1; ++o) {
                snprintf(buf, sizeof buf, "        Tl[%d] = u%d;\n", o, o);
                s += buf;
            }
            s += "        asm volatile(\"\" ::: \"memory\");\n";
            for (int o = g0; o < g1; ++o)
                if (seen[d + o]) {
                    snprintf(buf, sizeof buf, "        __hip_atomic_fetch_add(To + %d, -sgn * v%d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n", o, o);
                    s += buf;
                }
            s += "        asm volatile(\"\" ::: \"memory\");\n";
        }
        for (int o = g0; o < g1 && !lds_exchange; ++o) {
            if (seen[d + o])  // (both halves run this code: an output without B entries has V = 0 in both)
                snprintf(buf, sizeof buf, "        out(%d, sp_complete(u%d, v%d, sgn, half));\n", o, o, o);
            else
                snprintf(buf, sizeof buf, "        out(%d, u%d);\n", o, o);
            s += buf;
        }
        // a scheduling-region boundary: what the caller does with the group (LDS stores, dot products) is scheduled here, not
        // sunk to the end of the interval with every operand live until then
        s += "        __builtin_amdgcn_sched_barrier(0);\n    }\n";
    }
}
// Body of  double f(x, down, doth, mg, sgn):  this lane's part of <T^T x, D> = U . D_own - sgn V . D_other
//   = sum_g m_g (sum_{A entries} (+-) x[in] down[out] - sgn sum_{B entries} (+-) x[in] doth[out])
// -- one multiply-add per entry, no coefficient inside the sums
static inline void emit_dot(std::string &s, const std::vector<Term> &terms, int d, int n_mags) {
    char buf[256];
    std::vector<int> cu(n_mags, 0), cv(n_mags, 0);
    for (const Term &q : terms) (q.in >= d ? cv : cu)[q.mag]++;
    for (int g = 0; g < n_mags; ++g) {
        if (cu[g]) {
            snprintf(buf, sizeof buf, "    double eu%d = 0.0;\n", g);
            s += buf;
        }
        if (cv[g]) {
            snprintf(buf, sizeof buf, "    double ev%d = 0.0;\n", g);
            s += buf;
        }
    }
    for (const Term &q : terms) {
        const bool isv = q.in >= d;
        const int idx = isv ? q.in - d : q.in;
        if (!q.neg)
            snprintf(buf, sizeof buf, "    asm volatile(\"v_fmac_f64 %%0, %%1, %%2\" : \"+v\"(e%c%d) : \"v\"(x[%d]), \"v\"(%s[%d]));\n", isv ? 'v' : 'u', q.mag, idx, isv ? "doth" : "down", q.out);
        else
            snprintf(buf, sizeof buf, "    asm volatile(\"v_fma_f64 %%0, -%%1, %%2, %%0\" : \"+v\"(e%c%d) : \"v\"(x[%d]), \"v\"(%s[%d]));\n", isv ? 'v' : 'u', q.mag, idx, isv ? "doth" : "down", q.out);
        s += buf;
    }
    s += "    double r = 0.0;\n";
    for (int g = 0; g < n_mags; ++g) {
        if (cu[g] && cv[g])
            snprintf(buf, sizeof buf, "    r = __builtin_fma(mg.m%d, __builtin_fma(-sgn, ev%d, eu%d), r);\n", g, g, g);
        else if (cu[g])
            snprintf(buf, sizeof buf, "    r = __builtin_fma(mg.m%d, eu%d, r);\n", g, g);
        else if (cv[g])
            snprintf(buf, sizeof buf, "    r = __builtin_fma(mg.m%d, -sgn * ev%d, r);\n", g, g);
        else
            continue;
        s += buf;
    }
    s += "    return r;\n";
}
}  // namespace detail

// G0: n_g0 drifts (n*n column-major each), Gj: m drives.  The caller has verified the exact iso structure.
static inline SpPlan make_plan(int d, int m, const double *G0, int n_g0, const double *Gj) {
    using detail::Term;
    SpPlan P;
    P.d = d;
    P.m = m;
    P.n = 2 * d;
    const int n = P.n;
    const size_t nn = (size_t)n * n;
    std::vector<Term> terms;
    std::vector<int> rows, cols;
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < n; ++r) {
            const size_t pz = (size_t)r + (size_t)n * c;
            bool any = false;
            for (int b = 0; b < n_g0; ++b) any |= G0[b * nn + pz] != 0.0;
            for (int l = 0; l < m; ++l) any |= Gj[l * nn + pz] != 0.0;
            if (!any) continue;
            terms.push_back(detail::term_t(r, c, d, (int)rows.size()));
            rows.push_back(r);
            cols.push_back(c);
        }
    for (const Term &t : detail::emission_order(terms, d, kGroup)) {  // table order = emission order of sp_gt: the scalar loads walk it front to back
        const int r = rows[t.src], c = cols[t.src];
        P.row.push_back(r);
        P.col.push_back(c);
        P.pos.push_back(r + n * c);
        for (int l = 0; l < m; ++l) P.coef.push_back(Gj[l * nn + (size_t)r + (size_t)n * c]);
    }
    P.nz = (int)P.row.size();
    // padded so that the last chunk's loads (kChunk entries from any entry of the table) stay inside THIS interval's table: the
    // residual kernel's waves write their tables themselves, and a scalar load that strays into a neighbour's lines caches them
    // before their owner has written them (the scalar cache is not coherent: the owner then read stale coefficients at d = 31)
    P.nzp = (P.nz + kChunk + 15) & ~15;
    {
        std::vector<Term> tn;
        for (size_t k = 0; k < rows.size(); ++k) tn.push_back(detail::term_n(rows[k], cols[k], d, (int)k));
        for (const Term &t : detail::emission_order(tn, d, kGroup)) {
            const int r = rows[t.src], c = cols[t.src];
            P.row_n.push_back(r);
            P.col_n.push_back(c);
            P.pos_n.push_back(r + n * c);
            for (int l = 0; l < m; ++l) P.coef_n.push_back(Gj[l * nn + (size_t)r + (size_t)n * c]);
        }
    }
    // drives: the entries of the left column block; their distinct magnitudes (a handful for the reference's systems: ladder
    // operators) stay in scalar registers for the whole launch, the signs are instruction modifiers
    P.doff.assign(m + 1, 0);
    for (int l = 0; l < m; ++l) {
        P.doff[l] = (int)P.drow.size();
        for (int c = 0; c < d; ++c)
            for (int r = 0; r < n; ++r) {
                const double v = Gj[l * nn + (size_t)r + (size_t)n * c];
                if (v == 0.0) continue;
                const double mag = v < 0 ? -v : v;
                int gi = -1;
                for (size_t g = 0; g < P.mags.size(); ++g)
                    if (P.mags[g] == mag) gi = (int)g;
                if (gi < 0) {
                    gi = (int)P.mags.size();
                    P.mags.push_back(mag);
                }
                P.drow.push_back(r);
                P.dcol.push_back(c);
                P.dmagi.push_back(gi);
                P.dneg.push_back(v < 0);
            }
    }
    P.doff[m] = (int)P.drow.size();
    P.ok = P.mags.size() <= (size_t)kMaxMags;
    return P;
}

// Device functions shared by the pattern-compiled kernels (x = (a | b) is a column in the lane's convention; out(i, value)
// receives every top row of the result once, in groups of kGroup):
//   sp_gt(x, g, sgn, half, out)   G(u)^T x   (coefficients: the per-interval value table g, in ITS emission order)
//   sp_g(x, g, -sgn, half, out)   G(u) x     (its own table order: pos_n / coef_n)
//   sp_glt_<l>(a, b, mg, out)  G_l^T x      (coefficients: the resident magnitudes mg, signs as instruction modifiers)
//   sp_gltdot<l>(a, b, dq, mg) <G_l^T x, dq> over this half's rows
static inline std::string apply_functions(const SpPlan &P) {
    using detail::Term;
    const int d = P.d;
    std::string s;
    char buf[1600];
    snprintf(buf, sizeof buf,
             "#define SPD %d\n#define SPM %d\n#define SPN %d\n#define SPNZ %d\n#define SPNZP %d\n"
             "typedef const double __attribute__((address_space(4))) *sp_cptr;\n"
             "typedef double sp_v8d __attribute__((ext_vector_type(8)));\n"
             "typedef double sp_v4d __attribute__((ext_vector_type(4)));\n"
             "// u + sgn * (the v of the lane 32 positions away): v_permlane32_swap, VALU only\n"
             "static __device__ __forceinline__ double sp_complete(double u, double v, double sgn, int half) {\n"
             "    const int lo = __double2loint(v), hi = __double2hiint(v);\n"
             "    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);\n"
             "    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);\n"
             "    return __builtin_fma(sgn, __hiloint2double(half ? rh[0] : rh[1], half ? rl[0] : rl[1]), u);\n"
             "}\n",
             d, P.m, P.n, P.nz, P.nzp);
    s += buf;
    {
        std::vector<Term> t;  // the plan's entries ARE in emission order
        for (int k = 0; k < P.nz; ++k) t.push_back(detail::term_t(P.row[k], P.col[k], d, k));
        s += "template <class F> static __device__ __forceinline__ void sp_gt(const double (&x)[SPD], sp_cptr g, double sgn, int half, F out) {\n";
        detail::emit_groups(s, t, d, kGroup, "g", 0);
        s += "}\n";
    }
    {
        std::vector<Term> t;
        for (int k = 0; k < P.nz; ++k) t.push_back(detail::term_n(P.row_n[k], P.col_n[k], d, k));
        s += "template <class F> static __device__ __forceinline__ void sp_g(const double (&x)[SPD], sp_cptr g, double sgn, int half, F out) {\n";
        detail::emit_groups(s, t, d, kGroup, "g", 0);
        s += "}\n";
    }
    s += "struct sp_mags { double m0";
    for (size_t g = 1; g < std::max<size_t>(P.mags.size(), 1); ++g) {
        snprintf(buf, sizeof buf, ", m%zu", g);
        s += buf;
    }
    s += "; };\n#define SP_LOAD_MAGS(mg, tab) do {";
    for (size_t g = 0; g < std::max<size_t>(P.mags.size(), 1); ++g) {
        snprintf(buf, sizeof buf, " (mg).m%zu = (tab)[%zu];", g, g);
        s += buf;
    }
    s += " } while (0)\n";
    s += "template <int V> struct sp_ic { static constexpr int value = V; };\n";
    for (int l = 0; l < P.m; ++l) {
        std::vector<Term> tt;
        for (int k = P.doff[l]; k < P.doff[l + 1]; ++k) {
            Term q = detail::term_t(P.drow[k], P.dcol[k], d, k);
            q.neg = P.dneg[k] != 0;
            q.mag = P.dmagi[k];
            tt.push_back(q);
        }
        // G_l^T x -> the wave's LDS tile (Tl: this lane's rows, To: the other half's rows of the same column)
        snprintf(buf, sizeof buf, "static __device__ __forceinline__ void sp_glt_%d(const double (&x)[SPD], const sp_mags &mg, double sgn, double *Tl, double *To) {\n", l);
        s += buf;
        detail::emit_groups(s, detail::emission_order(tt, d, kGroup), d, kGroup, "mg", 0, true);
        s += "}\n";
        snprintf(buf, sizeof buf, "static __device__ __forceinline__ double sp_gltdot_%d(const double (&x)[SPD], const double (&down)[SPD], const double (&doth)[SPD], const sp_mags &mg, double sgn) {\n", l);
        s += buf;
        detail::emit_dot(s, tt, d, (int)P.mags.size());
        s += "}\n";
    }
    s += "template <int L> static __device__ __forceinline__ double sp_gltdot(const double (&x)[SPD], const double (&down)[SPD], const double (&doth)[SPD], const sp_mags &mg, double sgn) {\n";
    for (int l = 0; l < P.m; ++l) {
        snprintf(buf, sizeof buf, "    if constexpr (L == %d) return sp_gltdot_%d(x, down, doth, mg, sgn);\n", l, l);
        s += buf;
    }
    s += "    return 0.0;\n}\n";
    // the loader wave touches the next interval's value table (one 64-byte line per scalar load) so that the other waves' loads hit
    s += "#define SP_PREFETCH_G(ptr) do { sp_v8d pf_; asm volatile(\"";
    for (int c = 0; c * 8 < P.nzp; ++c) {
        snprintf(buf, sizeof buf, "s_load_dwordx16 %%0, %%1, %d\\n\\t", c * 64);
        s += buf;
    }
    s += "s_waitcnt lgkmcnt(0)\" : \"=&s\"(pf_) : \"s\"(ptr) : \"memory\"); } while (0)\n";
    // run-time (wave-uniform) dispatch for the two small drive-specific products of a drive wave; the cases exchange data with
    // the rest of the role through LDS only, so no register webs are merged behind the switch
    s += "#define SP_GLT_SWITCH(l, x, mg, sgn, Tl, To) switch (l) {";
    for (int l = 0; l < P.m; ++l) {
        snprintf(buf, sizeof buf, " case %d: sp_glt_%d(x, mg, sgn, Tl, To); break;", l, l);
        s += buf;
    }
    s += " default: break; }\n";
    return s;
}

}  // namespace pcl_codegen
