// pcl_codegen_v4.hpp -- host-only: source generator for the pattern-compiled FUSED residual + Jacobian kernel (any diagonal
// Pade order; DESIGN.md sections 4.2-4.4, device side: pcl_kernel_fused_sparse.hpp).
//
// Same idea as pcl_codegen.hpp (lane (half, c) owns its half of state column c; G(u) x is a straight line of multiply-adds whose
// register indices are the sparsity pattern), with two changes that remove every per-interval table from memory:
//   * G(u) = G0 + sum_l u_l G_l is never formed.  Drift entries take their coefficient from a LAUNCH-INVARIANT table (one per
//     ensemble member, in the emission order of the product, read with scalar loads in chunks of kV4Chunk); drive entries take
//     theirs from RESIDENT scalar registers  cf[k] = u_l |G_l entry|  (a handful of distinct magnitudes per system: ladder
//     operators), formed once per interval by every wave itself; the sign is an instruction modifier.
//   * the whole product  O = alpha Y + beta G(u) x  (x: registers; Y, O: LDS tiles [column][row]) is ONE asm statement: operand
//     loads, multiply-adds, the waits, the epilogue.  Nothing is in flight outside the statement and the compiler schedules
//     nothing inside it.  The two halves are completed in LDS: the lane writes alpha Y + beta U into its own slot and adds
//     -+ beta V into the other half's slot with ds_add_f64 (a wave's LDS operations complete in order).
#pragma once
#include <algorithm>

#include "pcl_codegen.hpp"

namespace pcl_codegen {

constexpr int kV4Group = 3;    // output rows accumulated together (2 * kV4Group independent chains; the product is ONE asm statement of
                               // 54 + 7 kV4Group + ~13 vector registers next to whatever the role keeps live: 128 per lane at 14 waves per CU)
constexpr int kV4Parts = 4;    // row ranges of the cooperative product (default; v4_parts() decides per system and order)
constexpr int kV4Chunk = 8;    // drift coefficients per scalar-load chunk (one s_load_dwordx16; two chunks of scalar registers in rotation)
constexpr int kV4MaxCf = 16;   // resident drive coefficients (drive, magnitude) the product keeps in scalar registers
constexpr int kV4MaxRes = 28;  // resident coefficients in all (scalar register pairs): the drives' first, then the drift's value classes by use
constexpr int kV4MaxResStreamed = 16;  // ... when some drift classes do not fit and the rest is streamed (the chunks take 32 scalar registers)

struct V4Term {
    int row;    // output row inside the half (0 .. d-1)
    bool isV;   // B block (accumulates V: completed by the OTHER half) or A block (U)
    int in;     // input index (state row of this half's x)
    int kind;   // 0: drift table entry `idx` (emission order) | 1: resident drive coefficient `idx` | 2: resident drift class `idx`
    int idx;
    bool neg;
};
struct V4GEnt {  // one entry of the left column block [A_l; B_l] of a drive, for the gather  (G_l w)_row
    int row, col;
    bool isB;
    int mag;
    bool neg;
};
struct V4Plan {
    int d = 0, m = 0, n = 0;
    std::vector<V4Term> terms;       // emission order
    std::vector<int> drift_pos;      // table order: column-major position r + n c in G0
    int n_drift_pad = 0;             // table length per member (padded: a chunk's loads never leave the member's table)
    std::vector<V4Term> terms_t;     // the same entries for G(u)^T x (the Hessian kernel): outputs by COLUMN of the left block
    std::vector<int> drift_pos_t;    // ... and the order its streamed drift entries are read in (a table of its own, same stride)
    std::vector<int> cf_l, cf_g;     // resident coefficient k = u[cf_l[k]] * mags[cf_g[k]]
    std::vector<double> mags;
    // The drift's entries fall into a few VALUE CLASSES (15 distinct magnitudes among the 91 entries of BASELINE config 3; 27 for
    // its perturbed ensemble members, the same classes in every member): the most used classes are resident too -- one table of
    // class values per member, read once per item -- and only the rest of the entries is streamed.
    int n_dcf = 0, n_dcf_pad = 0;    // resident drift classes; table stride per member
    std::vector<double> dcf_vals;    // [n_g0][n_dcf_pad]
    std::vector<char> hasU, hasV;    // per output row: any A / B entry
    std::vector<std::vector<V4GEnt>> gl;  // per drive
    bool ok = false;
};

static inline V4Plan make_v4_plan(int d, int m, const double *G0, int n_g0, const double *Gj) {
    V4Plan P;
    P.d = d;
    P.m = m;
    P.n = 2 * d;
    const int n = P.n;
    const size_t nn = (size_t)n * n;
    P.hasU.assign(d, 0);
    P.hasV.assign(d, 0);
    P.gl.resize(m);
    auto mag_index = [&](double v) {
        const double a = v < 0 ? -v : v;
        for (size_t g = 0; g < P.mags.size(); ++g)
            if (P.mags[g] == a) return (int)g;
        P.mags.push_back(a);
        return (int)P.mags.size() - 1;
    };
    auto cf_index = [&](int l, int g) {
        for (size_t k = 0; k < P.cf_l.size(); ++k)
            if (P.cf_l[k] == l && P.cf_g[k] == g) return (int)k;
        P.cf_l.push_back(l);
        P.cf_g.push_back(g);
        return (int)P.cf_l.size() - 1;
    };
    std::vector<std::vector<V4Term>> by(d);
    // drift: value classes over the members (key = the entry's values in every member, sign-normalised by the first nonzero one)
    std::vector<std::vector<double>> ckey;
    std::vector<int> ccount;
    std::vector<int> pos_class(nn, -1);
    std::vector<char> pos_neg(nn, 0);
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < n; ++r) {
            const size_t pz = (size_t)r + (size_t)n * c;
            std::vector<double> key(n_g0);
            double sg = 0.0;
            for (int b = 0; b < n_g0; ++b) {
                key[b] = G0[b * nn + pz];
                if (sg == 0.0 && key[b] != 0.0) sg = key[b] < 0 ? -1.0 : 1.0;
            }
            if (sg == 0.0) continue;
            for (double &v : key) v *= sg;
            int id = -1;
            for (size_t k = 0; k < ckey.size() && id < 0; ++k)
                if (ckey[k] == key) id = (int)k;
            if (id < 0) {
                id = (int)ckey.size();
                ckey.push_back(key);
                ccount.push_back(0);
            }
            ++ccount[id];
            pos_class[pz] = id;
            pos_neg[pz] = sg < 0;
        }
    for (int c = 0; c < d; ++c)  // the drives' coefficients first: they decide how many drift classes fit beside them
        for (int r = 0; r < n; ++r)
            for (int l = 0; l < m; ++l) {
                const double v = Gj[l * nn + (size_t)r + (size_t)n * c];
                if (v != 0.0) (void)cf_index(l, mag_index(v));
            }
    std::vector<int> class_res(ckey.size(), -1);
    {
        std::vector<int> order(ckey.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ccount[a] > ccount[b]; });
        // everything resident when it fits (no scalar load inside the product); otherwise a smaller resident set beside the chunk registers
        const bool all_fit = (int)order.size() + (int)P.cf_l.size() <= kV4MaxRes;
        const int room = std::max(0, (all_fit ? kV4MaxRes : kV4MaxResStreamed) - (int)P.cf_l.size());
        for (int k = 0; k < (int)order.size() && k < room; ++k) class_res[order[k]] = k;
        P.n_dcf = std::min<int>((int)order.size(), room);
        P.n_dcf_pad = (P.n_dcf + 8 + 7) & ~7;
        P.dcf_vals.assign((size_t)n_g0 * P.n_dcf_pad, 0.0);
        for (size_t k = 0; k < ckey.size(); ++k)
            if (class_res[k] >= 0)
                for (int b = 0; b < n_g0; ++b) P.dcf_vals[(size_t)b * P.n_dcf_pad + class_res[k]] = ckey[k][b];
    }
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < n; ++r) {
            const size_t pz = (size_t)r + (size_t)n * c;
            const int row = r < d ? r : r - d;
            const bool isV = r >= d;
            if (pos_class[pz] >= 0) {
                if (class_res[pos_class[pz]] >= 0)
                    by[row].push_back({row, isV, c, 2, class_res[pos_class[pz]], pos_neg[pz] != 0});
                else
                    by[row].push_back({row, isV, c, 0, (int)pz, false});  // idx: the position, replaced by the table index below
            }
            for (int l = 0; l < m; ++l) {
                const double v = Gj[l * nn + pz];
                if (v == 0.0) continue;
                const int g = mag_index(v);
                by[row].push_back({row, isV, c, 1, cf_index(l, g), v < 0});
                P.gl[l].push_back({row, c, isV, g, v < 0});
            }
        }
    // G(u)^T x: entry (r, c) of the left block feeds output row c from input row r (mod d); A block -> U, B block -> V
    // (top = U(0) + V(1), bottom = U(1) - V(0): the caller passes betas = -beta in half 0)
    std::vector<std::vector<V4Term>> by_t(d);
    for (int o = 0; o < d; ++o)
        for (const V4Term &t : by[o]) by_t[t.in].push_back({t.in, t.isV, t.row, t.kind, t.idx, t.neg});
    auto order_terms = [&](std::vector<std::vector<V4Term>> &lists, std::vector<V4Term> &out, std::vector<int> &dpos, bool mark) {
        for (int g0 = 0; g0 < d; g0 += kV4Group) {
            const int g1 = std::min(d, g0 + kV4Group);
            size_t longest = 0;
            for (int o = g0; o < g1; ++o) longest = std::max(longest, lists[o].size());
            for (size_t t = 0; t < longest; ++t)
                for (int o = g0; o < g1; ++o)
                    if (t < lists[o].size()) {
                        V4Term q = lists[o][t];
                        if (q.kind == 0) {
                            dpos.push_back(q.idx);
                            q.idx = (int)dpos.size() - 1;
                        }
                        if (mark) (q.isV ? P.hasV : P.hasU)[q.row] = 1;
                        out.push_back(q);
                    }
        }
    };
    order_terms(by, P.terms, P.drift_pos, true);
    order_terms(by_t, P.terms_t, P.drift_pos_t, false);
    P.n_drift_pad = (((int)P.drift_pos.size() + kV4Chunk - 1) / kV4Chunk + 1) * kV4Chunk + 4;
    P.n_drift_pad = (P.n_drift_pad + 7) & ~7;
    P.ok = (int)P.cf_l.size() <= kV4MaxCf && P.mags.size() <= (size_t)kMaxMags;
    return P;
}

// Host restatement of what the generated product computes (test hook: validates the term tables -- halves, signs, table order --
// against a dense product without a device).  x, y: full columns of length n; u: m controls; G0: one drift.
static inline void v4_reference_apply(const V4Plan &P, const double *G0, const double *Gj_unused, const double *u, const double *x, double *y) {
    (void)Gj_unused;
    const int d = P.d;
    std::vector<double> U0(d, 0.0), V0(d, 0.0), U1(d, 0.0), V1(d, 0.0);
    for (const V4Term &t : P.terms) {
        double c = t.kind == 0 ? G0[P.drift_pos[t.idx]] : (t.kind == 1 ? u[P.cf_l[t.idx]] * P.mags[P.cf_g[t.idx]] : P.dcf_vals[t.idx]);
        if (t.neg) c = -c;
        (t.isV ? V0 : U0)[t.row] += c * x[t.in];      // half 0 holds the top rows a
        (t.isV ? V1 : U1)[t.row] += c * x[d + t.in];  // half 1 the bottom rows b
    }
    for (int i = 0; i < d; ++i) {
        y[i] = U0[i] - V1[i];      // top = A a - B b
        y[d + i] = U1[i] + V0[i];  // bottom = A b + B a
    }
}

// ... and of the transposed product y = G(u)^T x (the Hessian kernel's)
static inline void v4_reference_apply_t(const V4Plan &P, const double *G0, const double *u, const double *x, double *y) {
    const int d = P.d;
    std::vector<double> U0(d, 0.0), V0(d, 0.0), U1(d, 0.0), V1(d, 0.0);
    for (const V4Term &t : P.terms_t) {
        double c = t.kind == 0 ? G0[P.drift_pos_t[t.idx]] : (t.kind == 1 ? u[P.cf_l[t.idx]] * P.mags[P.cf_g[t.idx]] : P.dcf_vals[t.idx]);
        if (t.neg) c = -c;
        (t.isV ? V0 : U0)[t.row] += c * x[t.in];
        (t.isV ? V1 : U1)[t.row] += c * x[d + t.in];
    }
    for (int i = 0; i < d; ++i) {
        y[i] = U0[i] + V1[i];      // top = A^T a + B^T b
        y[d + i] = U1[i] - V0[i];  // bottom = A^T b - B^T a
    }
}

namespace detail {
static inline std::string v4_chunk_reg(int chunk, int e) {
    char b[32];
    const int base = (chunk & 1 ? 52 : 36) + 2 * e;
    snprintf(b, sizeof b, "s[%d:%d]", base, base + 1);
    return b;
}
static inline void v4_emit_chunk_loads(std::string &s, int chunk) {
    char buf[200];
    snprintf(buf, sizeof buf, "        \"s_load_dwordx16 %s, %%[tab], %d\\n\\t\"\n", chunk & 1 ? "s[52:67]" : "s[36:51]", chunk * kV4Chunk * 8);
    s += buf;
}
}  // namespace detail

// The generated definitions: shape macros, the resident-coefficient struct, the product, the drives' gathers.
// entries per (drive, half) of the drives' transposed gathers' table (SP4_GT_TOTAL of the generated source: pcl_kernel_hess_cols.hpp): per output
// row the most terms any drive has there
static inline int v4_gather_total(const V4Plan &P) {
    std::vector<int> mx(P.d, 0);
    for (int l = 0; l < P.m; ++l) {
        std::vector<int> cnt(P.d, 0);
        for (const V4GEnt &e : P.gl[l]) mx[e.col] = std::max(mx[e.col], ++cnt[e.col]);
    }
    int tot = 0;
    for (int v : mx) tot += v;
    return tot;
}
// Row ranges (= waves) of the cooperative product of a system at order 2q: whole groups of kV4Group output rows, at most the m + 3 column waves
// that are idle while a workgroup's first powers are built (P, W, V, dW_l).  PCL_V4_PARTS overrides (measurements).
static inline int v4_parts(const V4Plan &P, int q) {
    (void)q;
    const int n_groups = (P.d + kV4Group - 1) / kV4Group;
    int want = kV4Parts;
    if (const char *e = getenv("PCL_V4_PARTS")) want = atoi(e);
    return std::max(1, std::min(std::min(want, P.m + 3), n_groups));
}
// np: LDS tiles the powers of G rotate through (>= 2 for q >= 2; q when they fit)
// variant: timing experiments of the product (WRONG results unless 0): 1 no ds_add_f64 | 2 no LDS operation in the epilogues | 3 one
// accumulator chain per output row group only half as deep (kV4Group rows -> plain v_mul of every term: no dependent chains)
static inline std::string v4_functions(const V4Plan &P, int q, int np, int variant_ = 0, bool with_hessian = false) {
    const int variant = variant_ & 7;            // (of the product)
    const int gdot_cols = (variant_ & 8) ? 7 : 32;  // columns of z per batch of the all-drive gather-dot: the whole half (nine at a time
                                                    // measured 7 % slower on the Hessian kernel: 199.9 vs 184.8 us per 8 trajectories, order 8)
    using detail::v4_chunk_reg;
    const int d = P.d, G = kV4Group;
    std::string s;
    char buf[512];
    snprintf(buf, sizeof buf,
             "#define SPD %d\n#define SPM %d\n#define SPN %d\n#define SP4Q %d\n#define SP4NP %d\n#define SP4NCF %d\n#define SP4NMAG %d\n#define SP4NDRIFT %d\n#define SP4NDCF %d\n#define SP4NDCFP %d\n"
             "typedef const double __attribute__((address_space(4))) *sp_cptr;\n",
             d, P.m, P.n, q, np, (int)std::max<size_t>(P.cf_l.size(), 1), (int)std::max<size_t>(P.mags.size(), 1), P.n_drift_pad, P.n_dcf, P.n_dcf_pad);
    s += buf;
    // resident coefficients.  (The products are VALU results in every lane; an "s" asm operand fed from a vector register sends
    // this compiler into an endless loop, so the value is moved to scalar registers explicitly.)
    s += "static __device__ __forceinline__ double sp4_uniform(double v) {\n"
         "    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));\n}\n";
    s += "struct sp4_cf { double c0";
    for (size_t k = 1; k < std::max<size_t>(P.cf_l.size(), 1); ++k) {
        snprintf(buf, sizeof buf, ", c%zu", k);
        s += buf;
    }
    for (int k = 0; k < P.n_dcf; ++k) {
        snprintf(buf, sizeof buf, ", g%d", k);
        s += buf;
    }
    s += "; };\n";
    s += "#define SP4_SET_DCF(cf, tab) do {";  // the member's drift class values (scalar loads)
    for (int k = 0; k < P.n_dcf; ++k) {
        snprintf(buf, sizeof buf, " (cf).g%d = (tab)[%d];", k, k);
        s += buf;
    }
    s += " } while (0)\n";
    s += "#define SP4_SET_CF(cf, u, mg) do {";
    for (size_t k = 0; k < P.cf_l.size(); ++k) {
        snprintf(buf, sizeof buf, " (cf).c%zu = sp4_uniform((u)[%d] * (mg)[%d]);", k, P.cf_l[k], P.cf_g[k]);
        s += buf;
    }
    if (P.cf_l.empty()) s += " (cf).c0 = 0.0;";
    s += " } while (0)\n";

    // ---- the product: O = alpha Y + beta G(u) x;  the variant without Y (alpha = 0: the powers of G) reads nothing in its
    //      epilogues and never waits for LDS ------------------------------------------------------------------------------------
    const int n_drift = (int)P.drift_pos.size();
    const int n_chunks = (n_drift + kV4Chunk - 1) / kV4Chunk;
    // [grp_lo, grp_hi): the output row groups this statement covers (a part of the product: other waves take the other rows)
    // unit_beta: every caller passes beta = 1 (the transposed product of the Hessian kernels): no multiply of the own half's rows, no scalar operand for it
    auto emit_product = [&](const char *name, bool with_y, bool transposed = false, int grp_lo = 0, int grp_hi = 1 << 20, bool unit_beta = false) {
        const std::vector<V4Term> &terms = transposed ? P.terms_t : P.terms;
        // Accumulators in two sets (group parity): a finished group is scaled in place (own value alpha Y + beta U in the U
        // register, betas V in the V register) and its LDS operations are issued BETWEEN the multiply-adds of the next group --
        // back to back behind their v_mul they cost a lone wave 20-27 cycles each (a third of the product).
        s += "// O[own] = alpha Y[own] + beta U,  O[other half] += betas V  (betas = -beta in half 1: top = A a - B b, bottom = A b + B a)\n";
        snprintf(buf, sizeof buf, "static __device__ __forceinline__ void %s(const double (&x)[SPD], unsigned vY, unsigned vO, unsigned vOo, double alpha, double beta, double betas, sp_cptr tab, const sp4_cf &cf) {\n", name);
        s += buf;
        s += "    double";
        for (int st = 0; st < 2; ++st)
            for (int g = 0; g < G; ++g) {
                snprintf(buf, sizeof buf, "%s aU%d_%d, aV%d_%d", (st || g) ? "," : "", st, g, st, g);
                s += buf;
            }
        if (with_y)
            for (int g = 0; g < G; ++g) {
                snprintf(buf, sizeof buf, ", yv%d", g);
                s += buf;
            }
        s += ";\n";
        if (!with_y) s += "    (void)vY;\n    (void)alpha;\n";
        if (unit_beta) s += "    (void)beta;  // (== 1 at every call)\n";
        if (n_chunks == 0) s += "    (void)tab;  // (every coefficient is resident)\n";
        s += "    asm volatile(\n";
        if (n_chunks > 0) detail::v4_emit_chunk_loads(s, 0);
        size_t ti = 0;
        std::vector<std::string> pending;  // LDS operations of the previous group
        size_t pend_i = 0;
        int ngroup = 0;
        for (int g0 = 0; g0 < d; g0 += G, ++ngroup) {
            const int g1 = std::min(d, g0 + G), st = ngroup & 1;
            if (ngroup < grp_lo || ngroup >= grp_hi) {  // (parts exist for fully resident coefficients only: no chunk bookkeeping to skip)
                while (ti < terms.size() && terms[ti].row >= g0 && terms[ti].row < g1) ++ti;
                continue;
            }
            if (with_y)  // this group's Y values
                for (int o = g0; o < g1; ++o) {
                    snprintf(buf, sizeof buf, "        \"ds_read_b64 %%[yv%d], %%[vY] offset:%d\\n\\t\"\n", o - g0, 8 * o);
                    s += buf;
                }
            size_t nterm = 0;
            for (size_t t = ti; t < terms.size() && terms[t].row >= g0 && terms[t].row < g1; ++t) ++nterm;
            const size_t npend = pending.size() - pend_i;
            const size_t every = npend ? std::max<size_t>(1, nterm / (npend + 1)) : 0;
            std::vector<char> seenU(G, 0), seenV(G, 0);
            size_t k = 0;
            for (; ti < terms.size() && terms[ti].row >= g0 && terms[ti].row < g1; ++ti, ++k) {
                const V4Term &t = terms[ti];
                if (every && k && k % every == 0 && pend_i < pending.size()) s += pending[pend_i++];
                std::string coef;
                if (t.kind == 0) {
                    const int chunk = t.idx / kV4Chunk, e = t.idx % kV4Chunk;
                    if (e == 0) {  // first use of a chunk: it has landed (requested one chunk ago); request the next one
                        s += "        \"s_waitcnt lgkmcnt(0)\\n\\t\"\n";
                        if (chunk + 1 < n_chunks) detail::v4_emit_chunk_loads(s, chunk + 1);
                    }
                    coef = v4_chunk_reg(chunk, e);
                } else {
                    snprintf(buf, sizeof buf, t.kind == 1 ? "%%[cf%d]" : "%%[dg%d]", t.idx);
                    coef = buf;
                }
                char acc[24];
                snprintf(acc, sizeof acc, "%%[a%c%d_%d]", t.isV ? 'V' : 'U', st, t.row - g0);
                char &sn = (t.isV ? seenV : seenU)[t.row - g0];
                if (!sn || variant == 3)
                    snprintf(buf, sizeof buf, "        \"v_mul_f64 %s, %s%s, %%[x%d]\\n\\t\"\n", acc, t.neg ? "-" : "", coef.c_str(), t.in);
                else if (!t.neg)
                    snprintf(buf, sizeof buf, "        \"v_fmac_f64 %s, %s, %%[x%d]\\n\\t\"\n", acc, coef.c_str(), t.in);
                else
                    snprintf(buf, sizeof buf, "        \"v_fma_f64 %s, -%s, %%[x%d], %s\\n\\t\"\n", acc, coef.c_str(), t.in, acc);
                sn = 1;
                s += buf;
            }
            while (pend_i < pending.size()) s += pending[pend_i++];
            pending.clear();
            pend_i = 0;
            // finish the group in place.  With Y: its values have landed -- a wave's LDS operations complete in order, so it is enough that
            // no more operations are in flight than were issued behind the Y reads (the previous group's writes and adds, `npend` of them:
            // waiting for those too cost a lone wave ~100 cycles per group, a third of the product; a scalar load in flight only makes
            // the wait longer, never shorter)
            if (with_y) {
                snprintf(buf, sizeof buf, "        \"s_waitcnt lgkmcnt(%zu)\\n\\t\"\n", std::min<size_t>(npend, 15));
                s += buf;
            }
            for (int o = g0; o < g1; ++o) {
                const int gi = o - g0;
                if (seenU[gi]) {
                    if (!(unit_beta && !with_y)) {
                        snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[aU%d_%d], %%[beta], %%[aU%d_%d]\\n\\t\"\n", st, gi, st, gi);
                        s += buf;
                    }
                    if (with_y) {
                        snprintf(buf, sizeof buf, "        \"v_fmac_f64 %%[aU%d_%d], %%[alpha], %%[yv%d]\\n\\t\"\n", st, gi, gi);
                        s += buf;
                    }
                } else {
                    if (with_y)
                        snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[aU%d_%d], %%[alpha], %%[yv%d]\\n\\t\"\n", st, gi, gi);
                    else
                        snprintf(buf, sizeof buf, "        \"v_mov_b64 %%[aU%d_%d], 0\\n\\t\"\n", st, gi);
                    s += buf;
                }
                snprintf(buf, sizeof buf, "        \"ds_write_b64 %%[vO], %%[aU%d_%d] offset:%d\\n\\t\"\n", st, gi, 8 * o);
                if (variant != 2) pending.push_back(buf);
                if (seenV[gi]) {
                    snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[aV%d_%d], %%[betas], %%[aV%d_%d]\\n\\t\"\n", st, gi, st, gi);
                    s += buf;
                    snprintf(buf, sizeof buf, "        \"ds_add_f64 %%[vOo], %%[aV%d_%d] offset:%d\\n\\t\"\n", st, gi, 8 * o);
                    if (variant != 1 && variant != 2) pending.push_back(buf);
                }
            }
        }
        for (const std::string &q_ : pending) s += q_;
        s += "        \"s_waitcnt lgkmcnt(0)\"\n        : ";
        for (int st = 0; st < 2; ++st)
            for (int g = 0; g < G; ++g) {
                snprintf(buf, sizeof buf, "%s[aU%d_%d] \"=&v\"(aU%d_%d), [aV%d_%d] \"=&v\"(aV%d_%d)", (st || g) ? ", " : "", st, g, st, g, st, g, st, g);
                s += buf;
            }
        if (with_y)
            for (int g = 0; g < G; ++g) {
                snprintf(buf, sizeof buf, ", [yv%d] \"=&v\"(yv%d)", g, g);
                s += buf;
            }
        s += "\n        : ";
        for (int i = 0; i < d; ++i) {
            snprintf(buf, sizeof buf, "%s[x%d] \"v\"(x[%d])", i ? ", " : "", i, i);
            s += buf;
        }
        if (with_y) s += ", [vY] \"v\"(vY), [alpha] \"s\"(alpha)";
        s += ", [vO] \"v\"(vO), [vOo] \"v\"(vOo), [betas] \"v\"(betas)";
        if (!(unit_beta && !with_y)) s += ", [beta] \"s\"(beta)";
        if (n_chunks > 0) s += ", [tab] \"s\"(tab)";
        for (size_t k = 0; k < std::max<size_t>(P.cf_l.size(), 1); ++k) {
            snprintf(buf, sizeof buf, ", [cf%zu] \"s\"(cf.c%zu)", k, k);
            s += buf;
        }
        for (int k = 0; k < P.n_dcf; ++k) {
            snprintf(buf, sizeof buf, ", [dg%d] \"s\"(cf.g%d)", k, k);
            s += buf;
        }
        s += "\n        : \"memory\"";
        for (int r = 36; r < 68 && n_chunks > 0; ++r) {  // the chunk registers of the streamed entries
            snprintf(buf, sizeof buf, ", \"s%d\"", r);
            s += buf;
        }
        s += ");\n}\n";
    };
    emit_product("sp4_product", true);
    emit_product("sp4_product0", false);
    // The product without Y in kV4Parts row ranges of about equal term counts (whole groups): the first item of a workgroup has its
    // powers of G built by the store-stream waves together -- they have nothing to store yet -- a quarter of the rows each.
    // Every row keeps the instruction sequence it has in sp4_product0: the same bits.
    const int n_groups = (d + G - 1) / G;
    const int nparts = v4_parts(P, q);
    const bool parts = n_chunks == 0 && n_groups >= nparts && nparts >= 2;
    snprintf(buf, sizeof buf, "#define SP4_COOP %d\n#define SP4_NPART %d\n", parts ? 1 : 0, nparts);
    s += buf;
    if (parts) {
        std::vector<std::pair<int, int>> part_rows;
        std::vector<size_t> cum(n_groups + 1, 0);
        for (const V4Term &t : P.terms) cum[t.row / G + 1]++;
        for (int g = 0; g < n_groups; ++g) cum[g + 1] += cum[g] + 4;  // (+ the group's epilogue)
        int lo = 0;
        for (int k = 0; k < nparts; ++k) {
            int hi = lo + 1;
            const size_t want = cum[n_groups] * (k + 1) / nparts;
            while (hi < n_groups - (nparts - 1 - k) && cum[hi] < want) ++hi;
            if (k == nparts - 1) hi = n_groups;
            const std::string part_name = "sp4_product0_p" + std::to_string(k);  // (buf is the emitter's scratch)
            emit_product(part_name.c_str(), false, false, lo, hi);
            const std::string party_name = "sp4_product_p" + std::to_string(k);  // ... and with Y: the cooperative residual kernel
            emit_product(party_name.c_str(), true, false, lo, hi);
            part_rows.push_back({lo * G, std::min(d, hi * G)});
            lo = hi;
        }
        s += "static __device__ __forceinline__ void sp4_product0_part(int part, const double (&x)[SPD], unsigned vO, unsigned vOo, double beta, double betas, sp_cptr tab, const sp4_cf &cf) {\n"
             "    switch (part) {\n";
        for (int k = 0; k < nparts; ++k) {
            snprintf(buf, sizeof buf, "    %s sp4_product0_p%d(x, 0u, vO, vOo, 0.0, beta, betas, tab, cf); break;\n", k == nparts - 1 ? "default:" : ("case " + std::to_string(k) + ":").c_str(), k);
            s += buf;
        }
        s += "    }\n}\n";
        s += "static __device__ __forceinline__ void sp4_product_part(int part, const double (&x)[SPD], unsigned vY, unsigned vO, unsigned vOo, double alpha, double beta, double betas, sp_cptr tab, const sp4_cf &cf) {\n"
             "    switch (part) {\n";
        for (int k = 0; k < nparts; ++k) {
            snprintf(buf, sizeof buf, "    %s sp4_product_p%d(x, vY, vO, vOo, alpha, beta, betas, tab, cf); break;\n", k == nparts - 1 ? "default:" : ("case " + std::to_string(k) + ":").c_str(), k);
            s += buf;
        }
        s += "    }\n}\n";
        s += "static __device__ __forceinline__ void sp4_part_rows(int part, int &r0, int &r1) {\n    switch (part) {\n";
        for (int k = 0; k < nparts; ++k) {
            snprintf(buf, sizeof buf, "    %s r0 = %d; r1 = %d; break;\n", k == nparts - 1 ? "default:" : ("case " + std::to_string(k) + ":").c_str(), part_rows[k].first, part_rows[k].second);
            s += buf;
        }
        s += "    }\n}\n";
    }
    if (with_hessian) {  // G(u)^T x for the Hessian of the Lagrangian: same statement shape, the transposed term tables
        emit_product("sp4_product_t", true, true);
        emit_product("sp4_product0_t", false, true, 0, 1 << 20, true);
    }

    // ---- the drives' gathers: X[own + i] = hs * (G_l w)_i for this lane's half-rows; Wo / Wx = this lane's own / other half of
    //      column c of w in LDS; sb = -1 in half 0, +1 in half 1 (top = A a - B b, bottom = A b + B a) ----------------------------
    for (int l = 0; l < P.m; ++l) {
        snprintf(buf, sizeof buf, "static __device__ __forceinline__ void sp4_gather_%d(const double *__restrict__ Wo, const double *__restrict__ Wx, double *__restrict__ X, double hs, double sb, const double (&mg)[SP4NMAG]) {\n", l);
        s += buf;
        // nine rows at a time: every read of a batch before its writes (the compiler cannot prove the tiles distinct), and never more
        // than nine results in registers next to the 27 of x the caller holds
        for (int i0 = 0; i0 < d; i0 += 9) {
            const int i1 = std::min(d, i0 + 9);
            s += "    {\n        double t_[9];\n";
            for (int i = i0; i < i1; ++i) {
                std::string ea, eb;
                for (const V4GEnt &e : P.gl[l])
                    if (e.row == i) {
                        std::string &dst = e.isB ? eb : ea;
                        snprintf(buf, sizeof buf, "%smg[%d] * %s[%d]", dst.empty() ? (e.neg ? "-" : "") : (e.neg ? " - " : " + "), e.mag, e.isB ? "Wx" : "Wo", e.col);
                        dst += buf;
                    }
                if (ea.empty() && eb.empty())
                    snprintf(buf, sizeof buf, "        t_[%d] = 0.0;\n", i - i0);
                else if (eb.empty())
                    snprintf(buf, sizeof buf, "        t_[%d] = hs * (%s);\n", i - i0, ea.c_str());
                else if (ea.empty())
                    snprintf(buf, sizeof buf, "        t_[%d] = hs * (sb * (%s));\n", i - i0, eb.c_str());
                else
                    snprintf(buf, sizeof buf, "        t_[%d] = hs * ((%s) + sb * (%s));\n", i - i0, ea.c_str(), eb.c_str());
                s += buf;
            }
            snprintf(buf, sizeof buf, "#pragma unroll\n        for (int i = 0; i < %d; ++i) X[%d + i] = t_[i];\n    }\n", i1 - i0, i0);
            s += buf;
        }
        s += "}\n";
    }
    if (with_hessian) {
        // X[own + c] = hs (G_l^T w)_c for this lane's half-rows: top = A^T a + B^T b, bottom = A^T b - B^T a (sb = +1 in half 0, -1 in half 1)
        for (int l = 0; l < P.m; ++l) {
            snprintf(buf, sizeof buf, "static __device__ __forceinline__ void sp4_gather_t_%d(const double *__restrict__ Wo, const double *__restrict__ Wx, double *__restrict__ X, double hs, double sb, const double (&mg)[SP4NMAG]) {\n", l);
            s += buf;
            for (int i0 = 0; i0 < d; i0 += 9) {
                const int i1 = std::min(d, i0 + 9);
                s += "    {\n        double t_[9];\n";
                for (int i = i0; i < i1; ++i) {
                    std::string ea, eb;
                    for (const V4GEnt &e : P.gl[l])
                        if (e.col == i) {  // transposed: output row = the entry's column, input = its row
                            std::string &dst = e.isB ? eb : ea;
                            snprintf(buf, sizeof buf, "%smg[%d] * %s[%d]", dst.empty() ? (e.neg ? "-" : "") : (e.neg ? " - " : " + "), e.mag, e.isB ? "Wx" : "Wo", e.row);
                            dst += buf;
                        }
                    if (ea.empty() && eb.empty())
                        snprintf(buf, sizeof buf, "        t_[%d] = 0.0;\n", i - i0);
                    else if (eb.empty())
                        snprintf(buf, sizeof buf, "        t_[%d] = hs * (%s);\n", i - i0, ea.c_str());
                    else if (ea.empty())
                        snprintf(buf, sizeof buf, "        t_[%d] = hs * (sb * (%s));\n", i - i0, eb.c_str());
                    else
                        snprintf(buf, sizeof buf, "        t_[%d] = hs * ((%s) + sb * (%s));\n", i - i0, ea.c_str(), eb.c_str());
                    s += buf;
                }
                snprintf(buf, sizeof buf, "#pragma unroll\n        for (int i = 0; i < %d; ++i) X[%d + i] = t_[i];\n    }\n", i1 - i0, i0);
                s += buf;
            }
            s += "}\n";
        }
        // The same gathers as ONE table for lanes that belong to different drives (pcl_kernel_hess_cols.hpp): entry [drive][half][row][term]
        // = (source row of the w column, 0 .. n-1) << 4 | coefficient index (0: none; 1 + 2 g: +mags[g]; 2 + 2 g: -mags[g]), the half's sign
        // of the B entries folded in (sb = +1 in half 0, -1 in half 1); sp4_gt_cnt(row) terms in row `row` (at most SP4_GTK), from entry sp4_gt_off(row).
        if (P.m > 0) {
            std::vector<std::vector<unsigned>> rows((size_t)P.m * 2 * d);
            size_t gtk = 1;
            for (int l = 0; l < P.m; ++l)
                for (int hf = 0; hf < 2; ++hf)
                    for (int i = 0; i < d; ++i) {
                        std::vector<unsigned> &r = rows[((size_t)l * 2 + hf) * d + i];
                        for (int pass = 0; pass < 2; ++pass)  // the A entries first, as the generated functions add them
                            for (const V4GEnt &e : P.gl[l])
                                if (e.col == i && e.isB == (pass == 1)) {
                                    const bool neg = e.neg != (e.isB && hf == 1);
                                    const unsigned src = (unsigned)((e.isB ? 1 - hf : hf) * d + e.row);
                                    r.push_back(src << 4 | (unsigned)(1 + 2 * e.mag + (neg ? 1 : 0)));
                                }
                        gtk = std::max(gtk, r.size());
                    }
            // row i takes cnt[i] entries in every (drive, half): the most any of them has there (the lanes of a wave run one instruction stream);
            // three 10-bit entries per dword, a whole number of dwords per (drive, half)
            std::vector<size_t> cnt(d, 0), off(d + 1, 0);
            for (size_t cls = 0; cls < (size_t)P.m * 2; ++cls)
                for (int i = 0; i < d; ++i) cnt[i] = std::max(cnt[i], rows[cls * d + i].size());
            for (int i = 0; i < d; ++i) off[i + 1] = off[i] + cnt[i];
            const size_t wpc = (off[d] + 2) / 3;
            snprintf(buf, sizeof buf, "#define SP4_GTK %zu\n#define SP4_GT_TOTAL %zu\n", gtk, off[d]);
            s += buf;
            s += "static __device__ constexpr int sp4_gt_cnt(int i) { constexpr int t_[SPD] = {";
            for (int i = 0; i < d; ++i) {
                snprintf(buf, sizeof buf, "%s%zu", i ? "," : "", cnt[i]);
                s += buf;
            }
            s += "}; return t_[i]; }\nstatic __device__ constexpr int sp4_gt_off(int i) { constexpr int t_[SPD + 1] = {";
            for (int i = 0; i <= d; ++i) {
                snprintf(buf, sizeof buf, "%s%zu", i ? "," : "", off[i]);
                s += buf;
            }
            snprintf(buf, sizeof buf, "}; return t_[i]; }\nstatic __device__ const unsigned sp4_gt_tab[%zu] = {", std::max<size_t>((size_t)P.m * 2 * wpc, 1));
            s += buf;
            for (size_t cls = 0; cls < (size_t)P.m * 2; ++cls) {
                std::vector<unsigned> words(wpc, 0u);
                for (int i = 0; i < d; ++i)
                    for (size_t k = 0; k < cnt[i]; ++k) {
                        const std::vector<unsigned> &r = rows[cls * d + i];
                        const size_t en = off[i] + k;
                        words[en / 3] |= (k < r.size() ? r[k] : 0u) << (10 * (en % 3));
                    }
                for (size_t w = 0; w < wpc; ++w) {
                    snprintf(buf, sizeof buf, "%s%uu", (cls || w) ? "," : "", words[w]);
                    s += buf;
                }
            }
            if ((size_t)P.m * 2 * wpc == 0) s += "0u";
            s += "};\n";
        }
        s += "#define SP4_GATHER_T_SWITCH(l, Wo, Wx, X, hs, sb, mg) switch (l) {";
        for (int l = 0; l < P.m; ++l) {
            snprintf(buf, sizeof buf, " case %d: sp4_gather_t_%d(Wo, Wx, X, hs, sb, mg); break;", l, l);
            s += buf;
        }
        s += " default: break; }\n";
    }
    if (with_hessian) {
        // out[l] = this lane's part of <v, G_l z> = sum_i v[i] (G_l z)_i over its half-rows, for EVERY drive at once (z: this lane's own / other
        // half of column c in LDS; sb = -1 in half 0, +1 in half 1 as in sp4_gather_<l>).  The column of z is read once per half into registers
        // (54 LDS reads instead of 54 per drive: six drive waves doing six gather-dots each kept the LDS pipe busy for most of an
        // interval), the entries of a drive are summed per magnitude (one multiply-add per entry, the magnitudes applied at the end),
        // the drives' chains interleaved.
        {
            s += "static __device__ __forceinline__ void sp4_gdot_all(const double *__restrict__ Zo, const double *__restrict__ Zx, const double (&v)[SPD], double sb, const double (&mg)[SP4NMAG], double (&out)[SPM > 0 ? SPM : 1]) {\n";
            std::vector<std::vector<std::string>> fin(P.m);  // per drive: the final sum's terms
            for (int part = 0; part < 2; ++part) {
                bool any = false;
                for (int l = 0; l < P.m; ++l)
                    for (const V4GEnt &e : P.gl[l]) any = any || (e.isB == (part == 1));
                if (!any) continue;
                s += "    {\n";
                // accumulators per (drive, magnitude); the entries of the drives interleaved
                std::vector<std::vector<const V4GEnt *>> lists(P.m);
                std::vector<std::vector<int>> mags_of(P.m);
                for (int l = 0; l < P.m; ++l) {
                    for (const V4GEnt &e : P.gl[l])
                        if (e.isB == (part == 1)) {
                            lists[l].push_back(&e);
                            if (std::find(mags_of[l].begin(), mags_of[l].end(), e.mag) == mags_of[l].end()) mags_of[l].push_back(e.mag);
                        }
                    for (int g : mags_of[l]) {
                        snprintf(buf, sizeof buf, "        double %c%d_%d = 0.0;\n", part ? 'b' : 'a', l, g);
                        s += buf;
                    }
                }
                for (int c0 = 0; c0 < d; c0 += gdot_cols) {
                    const int c1 = std::min(d, c0 + gdot_cols);
                    snprintf(buf, sizeof buf, "        {\n            double z_[32];\n#pragma unroll\n            for (int i = 0; i < %d; ++i) z_[i] = %s[%d + i];\n", c1 - c0, part ? "Zx" : "Zo", c0);
                    s += buf;
                    std::vector<size_t> pos(P.m, 0);
                    bool more = true;
                    while (more) {
                        more = false;
                        for (int l = 0; l < P.m; ++l) {
                            while (pos[l] < lists[l].size() && !(lists[l][pos[l]]->col >= c0 && lists[l][pos[l]]->col < c1)) ++pos[l];
                            if (pos[l] < lists[l].size()) {
                                const V4GEnt &e = *lists[l][pos[l]++];
                                snprintf(buf, sizeof buf, "            %c%d_%d = __builtin_fma(%sv[%d], z_[%d], %c%d_%d);\n", part ? 'b' : 'a', l, e.mag, e.neg ? "-" : "", e.row, e.col - c0, part ? 'b' : 'a', l, e.mag);
                                s += buf;
                                more = true;
                            }
                        }
                    }
                    s += "            asm volatile(\"\" ::: \"memory\");\n        }\n";
                }
                for (int l = 0; l < P.m; ++l) {
                    if (mags_of[l].empty()) continue;
                    std::string sum;
                    for (int g : mags_of[l]) {
                        snprintf(buf, sizeof buf, "%smg[%d] * %c%d_%d", sum.empty() ? "" : " + ", g, part ? 'b' : 'a', l, g);
                        sum += buf;
                    }
                    snprintf(buf, sizeof buf, "        const double p%d_%d = %s;\n", part, l, sum.c_str());
                    s += buf;
                    snprintf(buf, sizeof buf, part ? "sb * p%d_%d" : "p%d_%d", part, l);
                    fin[l].push_back(buf);
                }
                // (the results leave the block through out[]; a memory clobber keeps the other half's reads behind this half's arithmetic)
                for (int l = 0; l < P.m; ++l)
                    if (!mags_of[l].empty()) {
                        snprintf(buf, sizeof buf, "        out[%d] %s %s;\n", l, (part == 1 && fin[l].size() == 2) ? "+=" : "=", fin[l].back().c_str());
                        s += buf;
                    }
                s += "        asm volatile(\"\" ::: \"memory\");\n    }\n";
            }
            for (int l = 0; l < P.m; ++l)
                if (fin[l].empty()) {
                    snprintf(buf, sizeof buf, "    out[%d] = 0.0;\n", l);
                    s += buf;
                }
            s += "    (void)Zo; (void)Zx; (void)v; (void)sb; (void)mg;\n}\n";
        }
    }
    s += "#define SP4_GATHER_SWITCH(l, Wo, Wx, X, hs, sb, mg) switch (l) {";
    for (int l = 0; l < P.m; ++l) {
        snprintf(buf, sizeof buf, " case %d: sp4_gather_%d(Wo, Wx, X, hs, sb, mg); break;", l, l);
        s += buf;
    }
    s += " default: break; }\n";
    return s;
}

}  // namespace pcl_codegen
