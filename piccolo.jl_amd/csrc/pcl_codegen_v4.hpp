// pcl_codegen_v4.hpp -- host-only: source generator for the pattern-compiled FUSED residual + Jacobian kernel (any diagonal
// Pade order; DESIGN.md section 4.9, device side: pcl_kernel_fused_sparse.hpp).
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

#include "pcl_codegen.hpp"

namespace pcl_codegen {

constexpr int kV4Group = 4;    // output rows accumulated together (2 * kV4Group independent chains; the product is ONE asm statement of
                               // 54 + 7 kV4Group + ~13 vector registers next to whatever the role keeps live: 128 per lane at 14 waves per CU)
constexpr int kV4Chunk = 8;    // drift coefficients per scalar-load chunk (one s_load_dwordx16; two chunks of scalar registers in rotation)
constexpr int kV4MaxCf = 16;   // resident drive coefficients (drive, magnitude) the product keeps in scalar registers

struct V4Term {
    int row;    // output row inside the half (0 .. d-1)
    bool isV;   // B block (accumulates V: completed by the OTHER half) or A block (U)
    int in;     // input index (state row of this half's x)
    int kind;   // 0: drift table entry `idx` (emission order) | 1: resident coefficient `idx`
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
    std::vector<int> cf_l, cf_g;     // resident coefficient k = u[cf_l[k]] * mags[cf_g[k]]
    std::vector<double> mags;
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
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < n; ++r) {
            const size_t pz = (size_t)r + (size_t)n * c;
            bool drift = false;
            for (int b = 0; b < n_g0; ++b) drift |= G0[b * nn + pz] != 0.0;
            const int row = r < d ? r : r - d;
            const bool isV = r >= d;
            if (drift) by[row].push_back({row, isV, c, 0, (int)pz, false});  // idx: the position, replaced by the table index below
            for (int l = 0; l < m; ++l) {
                const double v = Gj[l * nn + pz];
                if (v == 0.0) continue;
                const int g = mag_index(v);
                by[row].push_back({row, isV, c, 1, cf_index(l, g), v < 0});
                P.gl[l].push_back({row, c, isV, g, v < 0});
            }
        }
    for (int g0 = 0; g0 < d; g0 += kV4Group) {
        const int g1 = std::min(d, g0 + kV4Group);
        size_t longest = 0;
        for (int o = g0; o < g1; ++o) longest = std::max(longest, by[o].size());
        for (size_t t = 0; t < longest; ++t)
            for (int o = g0; o < g1; ++o)
                if (t < by[o].size()) {
                    V4Term q = by[o][t];
                    if (q.kind == 0) {
                        P.drift_pos.push_back(q.idx);
                        q.idx = (int)P.drift_pos.size() - 1;
                    }
                    (q.isV ? P.hasV : P.hasU)[q.row] = 1;
                    P.terms.push_back(q);
                }
    }
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
        double c = t.kind == 0 ? G0[P.drift_pos[t.idx]] : u[P.cf_l[t.idx]] * P.mags[P.cf_g[t.idx]];
        if (t.neg) c = -c;
        (t.isV ? V0 : U0)[t.row] += c * x[t.in];      // half 0 holds the top rows a
        (t.isV ? V1 : U1)[t.row] += c * x[d + t.in];  // half 1 the bottom rows b
    }
    for (int i = 0; i < d; ++i) {
        y[i] = U0[i] - V1[i];      // top = A a - B b
        y[d + i] = U1[i] + V0[i];  // bottom = A b + B a
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
static inline std::string v4_functions(const V4Plan &P, int q) {
    using detail::v4_chunk_reg;
    const int d = P.d, G = kV4Group;
    std::string s;
    char buf[512];
    snprintf(buf, sizeof buf,
             "#define SPD %d\n#define SPM %d\n#define SPN %d\n#define SP4Q %d\n#define SP4NCF %d\n#define SP4NMAG %d\n#define SP4NDRIFT %d\n"
             "typedef const double __attribute__((address_space(4))) *sp_cptr;\n",
             d, P.m, P.n, q, (int)std::max<size_t>(P.cf_l.size(), 1), (int)std::max<size_t>(P.mags.size(), 1), P.n_drift_pad);
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
    s += "; };\n";
    s += "#define SP4_SET_CF(cf, u, mg) do {";
    for (size_t k = 0; k < P.cf_l.size(); ++k) {
        snprintf(buf, sizeof buf, " (cf).c%zu = sp4_uniform((u)[%d] * (mg)[%d]);", k, P.cf_l[k], P.cf_g[k]);
        s += buf;
    }
    if (P.cf_l.empty()) s += " (cf).c0 = 0.0;";
    s += " } while (0)\n";

    // ---- the product: O = alpha Y + beta G(u) x -----------------------------------------------------------------------------
    s += "// O[own] = alpha Y[own] + beta U,  O[other half] += betas V  (betas = -beta in half 1: top = A a - B b, bottom = A b + B a)\n";
    s += "static __device__ __forceinline__ void sp4_product(const double (&x)[SPD], unsigned vY, unsigned vO, unsigned vOo, double alpha, double beta, double betas, sp_cptr tab, const sp4_cf &cf) {\n";
    s += "    double";
    for (int g = 0; g < G; ++g) {
        snprintf(buf, sizeof buf, "%s aU%d, aV%d, yv%d", g ? "," : "", g, g, g);
        s += buf;
    }
    s += ", t0, t1;\n";
    s += "    asm volatile(\n";
    const int n_drift = (int)P.drift_pos.size();
    const int n_chunks = (n_drift + kV4Chunk - 1) / kV4Chunk;
    if (n_chunks > 0) detail::v4_emit_chunk_loads(s, 0);
    size_t ti = 0;
    int tsel = 0;
    for (int g0 = 0; g0 < d; g0 += G) {
        const int g1 = std::min(d, g0 + G);
        // this group's Y values
        for (int o = g0; o < g1; ++o) {
            snprintf(buf, sizeof buf, "        \"ds_read_b64 %%[yv%d], %%[vY] offset:%d\\n\\t\"\n", o - g0, 8 * o);
            s += buf;
        }
        std::vector<char> seenU(G, 0), seenV(G, 0);
        for (; ti < P.terms.size() && P.terms[ti].row >= g0 && P.terms[ti].row < g1; ++ti) {
            const V4Term &t = P.terms[ti];
            std::string coef;
            if (t.kind == 0) {
                const int chunk = t.idx / kV4Chunk, e = t.idx % kV4Chunk;
                if (e == 0) {  // first use of a chunk: it has landed (requested one chunk ago); request the next one
                    s += "        \"s_waitcnt lgkmcnt(0)\\n\\t\"\n";
                    if (chunk + 1 < n_chunks) detail::v4_emit_chunk_loads(s, chunk + 1);
                }
                coef = v4_chunk_reg(chunk, e);
            } else {
                snprintf(buf, sizeof buf, "%%[cf%d]", t.idx);
                coef = buf;
            }
            char acc[24];
            snprintf(acc, sizeof acc, "%%[a%c%d]", t.isV ? 'V' : 'U', t.row - g0);
            char &sn = (t.isV ? seenV : seenU)[t.row - g0];
            if (!sn)
                snprintf(buf, sizeof buf, "        \"v_mul_f64 %s, %s%s, %%[x%d]\\n\\t\"\n", acc, t.neg ? "-" : "", coef.c_str(), t.in);
            else if (!t.neg)
                snprintf(buf, sizeof buf, "        \"v_fmac_f64 %s, %s, %%[x%d]\\n\\t\"\n", acc, coef.c_str(), t.in);
            else
                snprintf(buf, sizeof buf, "        \"v_fma_f64 %s, -%s, %%[x%d], %s\\n\\t\"\n", acc, coef.c_str(), t.in, acc);
            sn = 1;
            s += buf;
        }
        // epilogue of the group: the Y values (and whatever else this wave has in flight) have landed
        s += "        \"s_waitcnt lgkmcnt(0)\\n\\t\"\n";
        for (int o = g0; o < g1; ++o) {
            const int gi = o - g0;
            const char *ta = "t0", *tb = "t1";
            (void)tsel;
            if (seenU[gi])
                snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[%s], %%[beta], %%[aU%d]\\n\\t\"\n        \"v_fmac_f64 %%[%s], %%[alpha], %%[yv%d]\\n\\t\"\n", ta, gi, ta, gi);
            else
                snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[%s], %%[alpha], %%[yv%d]\\n\\t\"\n", ta, gi);
            s += buf;
            snprintf(buf, sizeof buf, "        \"ds_write_b64 %%[vO], %%[%s] offset:%d\\n\\t\"\n", ta, 8 * o);
            s += buf;
            if (seenV[gi]) {
                snprintf(buf, sizeof buf, "        \"v_mul_f64 %%[%s], %%[betas], %%[aV%d]\\n\\t\"\n        \"ds_add_f64 %%[vOo], %%[%s] offset:%d\\n\\t\"\n", tb, gi, tb, 8 * o);
                s += buf;
            }
        }
    }
    s += "        \"s_waitcnt lgkmcnt(0)\"\n        : ";
    for (int g = 0; g < G; ++g) {
        snprintf(buf, sizeof buf, "%s[aU%d] \"=&v\"(aU%d), [aV%d] \"=&v\"(aV%d), [yv%d] \"=&v\"(yv%d)", g ? ", " : "", g, g, g, g, g, g);
        s += buf;
    }
    s += ", [t0] \"=&v\"(t0), [t1] \"=&v\"(t1)\n        : ";
    for (int i = 0; i < d; ++i) {
        snprintf(buf, sizeof buf, "%s[x%d] \"v\"(x[%d])", i ? ", " : "", i, i);
        s += buf;
    }
    s += ", [vY] \"v\"(vY), [vO] \"v\"(vO), [vOo] \"v\"(vOo), [alpha] \"v\"(alpha), [beta] \"v\"(beta), [betas] \"v\"(betas), [tab] \"s\"(tab)";
    for (size_t k = 0; k < std::max<size_t>(P.cf_l.size(), 1); ++k) {
        snprintf(buf, sizeof buf, ", [cf%zu] \"s\"(cf.c%zu)", k, k);
        s += buf;
    }
    s += "\n        : \"memory\"";
    for (int r = 36; r < 68; ++r) {
        snprintf(buf, sizeof buf, ", \"s%d\"", r);
        s += buf;
    }
    s += ");\n}\n";

    // ---- the drives' gathers: X[own + i] = hs * (G_l w)_i for this lane's half-rows; Wo / Wx = this lane's own / other half of
    //      column c of w in LDS; sb = -1 in half 0, +1 in half 1 (top = A a - B b, bottom = A b + B a) ----------------------------
    for (int l = 0; l < P.m; ++l) {
        snprintf(buf, sizeof buf, "static __device__ __forceinline__ void sp4_gather_%d(const double *__restrict__ Wo, const double *__restrict__ Wx, double *__restrict__ X, double hs, double sb, const double (&mg)[SP4NMAG]) {\n", l);
        s += buf;
        s += "    double t_[SPD];\n";
        for (int i = 0; i < d; ++i) {
            std::string ea, eb;
            for (const V4GEnt &e : P.gl[l])
                if (e.row == i) {
                    std::string &dst = e.isB ? eb : ea;
                    snprintf(buf, sizeof buf, "%smg[%d] * %s[%d]", dst.empty() ? (e.neg ? "-" : "") : (e.neg ? " - " : " + "), e.mag, e.isB ? "Wx" : "Wo", e.col);
                    dst += buf;
                }
            if (ea.empty() && eb.empty())
                snprintf(buf, sizeof buf, "    t_[%d] = 0.0;\n", i);
            else if (eb.empty())
                snprintf(buf, sizeof buf, "    t_[%d] = hs * (%s);\n", i, ea.c_str());
            else if (ea.empty())
                snprintf(buf, sizeof buf, "    t_[%d] = hs * (sb * (%s));\n", i, eb.c_str());
            else
                snprintf(buf, sizeof buf, "    t_[%d] = hs * ((%s) + sb * (%s));\n", i, ea.c_str(), eb.c_str());
            s += buf;
        }
        s += "#pragma unroll\n    for (int i = 0; i < SPD; ++i) X[i] = t_[i];\n";
        s += "}\n";
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
