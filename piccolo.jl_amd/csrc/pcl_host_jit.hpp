// pcl_host_jit.hpp -- part of piccolo_hip.hip (included there, in place): run-time compilation of the pattern-compiled kernels with hiprtc, the
// persistent code objects (content-hashed modules in <library dir>/prebuilt and in the user's cache), pcl_jit_prebuild and the source inspection hooks.
#pragma once
// --- run-time shape specialisation (hiprtc) -----------------------------------------------------------------------
// The wave-synchronous kernels are 1.3-3x faster with compile-time Hilbert dimension / drive count (constant LDS strides,
// no SGPR spills).  A few shapes are instantiated statically; any other shape is compiled on first use from the kernel
// headers that sit next to this library (pcl_*.hpp, located with dladdr) -- about 1.5 s, cached for the process.
// libhiprtc is opened lazily; when it or the headers are missing the run-time-shape instances are used (same results).
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
namespace {
struct JitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
};
struct HiprtcApi {
    void *h = nullptr;
    int (*CreateProgram)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*AddNameExpression)(void *, const char *) = nullptr;
    int (*CompileProgram)(void *, int, const char **) = nullptr;
    int (*GetLoweredName)(void *, const char *, const char **) = nullptr;
    int (*GetCodeSize)(void *, size_t *) = nullptr;
    int (*GetCode)(void *, char *) = nullptr;
    int (*GetProgramLogSize)(void *, size_t *) = nullptr;
    int (*GetProgramLog)(void *, char *) = nullptr;
    int (*DestroyProgram)(void **) = nullptr;
    int (*Version)(int *, int *) = nullptr;
};
std::mutex g_jit_mutex;
std::map<std::string, JitKernel> g_jit;  // key: device | template instance
HiprtcApi g_rtc;
int64_t g_jit_compiles = 0, g_jit_cache_hits = 0, g_jit_fallbacks = 0;
std::string g_jit_note;

// ---- persistent code objects --------------------------------------------------------------------------------------------------
// A compiled module is kept on disk under the hash of everything it was compiled from (generated source, the kernel headers it
// includes, the compiler options, the hiprtc version): <library dir>/prebuilt/<hash>.hsaco (written by pcl_jit_prebuild -- what
// __graft_entry__.build() fills for the BASELINE configs; travels with the library) is looked at first, then the user's cache
// ($PCL_JIT_CACHE_DIR, else $XDG_CACHE_HOME/piccolo_hip, else ~/.cache/piccolo_hip), which every run-time compilation also writes
// (temporary file + rename: ranks of one job may compile the same module at the same time).  PCL_JIT_CACHE=0 switches both off.
// The user's cache is only used when it is the user's: no HOME / XDG_CACHE_HOME / PCL_JIT_CACHE_DIR means no user cache (never a shared /tmp
// path), the directory is created 0700 and must belong to this user and be closed to group and others, a file must belong to this user;
// every file carries its payload's length and a checksum, and one that does not unpack or load is removed and compiled afresh.
struct Hash128 {
    uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
    void add(const void *p_, size_t n) {
        const unsigned char *p = (const unsigned char *)p_;
        for (size_t i = 0; i < n; ++i) {
            a = (a ^ p[i]) * 0x100000001b3ull;
            b = (b ^ p[i]) * 0x9e3779b97f4a7c15ull + (b >> 29);
        }
    }
    void add(const std::string &x) {
        add(x.data(), x.size());
        const unsigned char z = 0;
        add(&z, 1);
    }
    std::string hex() const {
        char buf[40];
        snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
        return buf;
    }
};
bool cache_enabled() {
    const char *e = getenv("PCL_JIT_CACHE");
    return !(e && e[0] == '0');
}
std::string user_cache_dir() {
    if (const char *e = getenv("PCL_JIT_CACHE_DIR")) return e;
    if (const char *x = getenv("XDG_CACHE_HOME"))
        if (x[0]) return std::string(x) + "/piccolo_hip";
    if (const char *h = getenv("HOME"))
        if (h[0]) return std::string(h) + "/.cache/piccolo_hip";
    return std::string();  // nowhere that is this user's: no user cache (a predictable world-writable path could be pre-planted)
}
// the directory is this user's and nobody else can write to it; `file` (when given) inside it is this user's too
bool user_cache_trusted(const std::string &dir, const std::string &file = std::string()) {
    struct stat st;
    if (dir.empty() || stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;
    if (file.empty()) return true;
    return lstat((dir + "/" + file).c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid();
}
bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear();
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return !out.empty();
}
void mkdir_p(const std::string &dir);
bool write_file_atomic(const std::string &dir, const std::string &name, const std::vector<char> &data) {
    mkdir_p(dir);
    char tmpl[64];
    snprintf(tmpl, sizeof tmpl, ".tmp.%ld.%p", (long)getpid(), (const void *)&data);
    const std::string tmp = dir + "/" + name + tmpl, fin = dir + "/" + name;
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    ok = (fclose(f) == 0) && ok;  // (a short write may only show at close: the file must be whole before it gets its name)
    if (!ok || rename(tmp.c_str(), fin.c_str()) != 0) {
        remove(tmp.c_str());
        return false;
    }
    return true;
}

bool rtc_load() {
    if (g_rtc.h) return true;
    void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libhiprtc.so.7", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        g_jit_note = std::string("dlopen(libhiprtc.so): ") + dlerror();
        return false;
    }
    HiprtcApi a;
    a.h = h;
#define RTC_SYM(field, name) a.field = (decltype(a.field))dlsym(h, name)
    RTC_SYM(CreateProgram, "hiprtcCreateProgram");
    RTC_SYM(AddNameExpression, "hiprtcAddNameExpression");
    RTC_SYM(CompileProgram, "hiprtcCompileProgram");
    RTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
    RTC_SYM(GetCodeSize, "hiprtcGetCodeSize");
    RTC_SYM(GetCode, "hiprtcGetCode");
    RTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
    RTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
    RTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
    RTC_SYM(Version, "hiprtcVersion");
#undef RTC_SYM
    if (!a.CreateProgram || !a.AddNameExpression || !a.CompileProgram || !a.GetLoweredName || !a.GetCodeSize || !a.GetCode || !a.DestroyProgram) {
        g_jit_note = "libhiprtc lacks the expected symbols";
        return false;
    }
    g_rtc = a;
    return true;
}

bool slurp(const std::string &path, std::string &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return !out.empty();
}


void mkdir_p(const std::string &dir) {  // (the leaf -- the cache itself -- for this user only)
    for (size_t i = 1; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), i == dir.size() ? 0700 : 0755);
}
#ifdef PCL_PROFILE
static const char *const kJitOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-DPCL_PROFILE"};
#else
static const char *const kJitOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
#endif
std::string jit_cache_key(const std::string &source, const std::string *hdr, int nh, const char *name_expr) {
    Hash128 h;
    h.add(std::string("pcl-jit-2"));
    h.add(source);
    for (int i = 0; i < nh; ++i) h.add(hdr[i]);
    for (const char *o : kJitOpts) h.add(std::string(o));
    if (const char *e = getenv("PCL_JIT_OPTS")) h.add(std::string(e));
    h.add(std::string(name_expr ? name_expr : ""));
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);  // the ROCm release (hiprtc ships with it); asked of the runtime that is loaded anyway -- a cache hit never opens libhiprtc
    h.add(std::to_string(ver));
    return h.hex();
}
// on disk: "PCL2" | uint32 length of the kernel's (lowered) name | uint64 length of the code object | uint64 checksum of name + code | name | code object
static uint64_t module_checksum(const char *name, size_t nn, const char *code, size_t nc) {
    Hash128 h;
    h.add(name, nn);
    h.add(code, nc);
    return h.a ^ (h.b << 1);
}
std::vector<char> pack_module(const std::vector<char> &code, const std::string &lname) {
    std::vector<char> out;
    const uint32_t n = (uint32_t)lname.size();
    const uint64_t nc = (uint64_t)code.size(), ck = module_checksum(lname.data(), lname.size(), code.data(), code.size());
    out.insert(out.end(), {'P', 'C', 'L', '2'});
    out.insert(out.end(), (const char *)&n, (const char *)&n + 4);
    out.insert(out.end(), (const char *)&nc, (const char *)&nc + 8);
    out.insert(out.end(), (const char *)&ck, (const char *)&ck + 8);
    out.insert(out.end(), lname.begin(), lname.end());
    out.insert(out.end(), code.begin(), code.end());
    return out;
}
bool unpack_module(const std::vector<char> &blob, std::vector<char> &code, std::string &lname, bool plain_name) {
    if (blob.size() < 24 || memcmp(blob.data(), "PCL2", 4) != 0) return false;
    uint32_t n;
    uint64_t nc, ck;
    memcpy(&n, blob.data() + 4, 4);
    memcpy(&nc, blob.data() + 8, 8);
    memcpy(&ck, blob.data() + 16, 8);
    if (nc < 16 || (uint64_t)blob.size() != 24 + (uint64_t)n + nc) return false;  // truncated or padded
    if (module_checksum(blob.data() + 24, n, blob.data() + 24 + n, (size_t)nc) != ck) return false;
    if (!plain_name) lname.assign(blob.data() + 24, n);
    code.assign(blob.begin() + 24 + n, blob.end());
    return true;
}
bool rtc_compile(const std::string &source, const char **hdrp, const char *const *names, int nh, const char *name_expr, bool plain_name, const std::string &what,
                 std::vector<char> &code, std::string &lname) {
    void *prog = nullptr;
    if (g_rtc.CreateProgram(&prog, source.c_str(), "pcl_jit.hip", nh, hdrp, (const char **)names) != 0) {
        g_jit_note = "hiprtcCreateProgram failed";
        return false;
    }
    if (!plain_name) g_rtc.AddNameExpression(prog, name_expr);
    // per-module compiler options: the leading lines `// pcl-jit-opt: <tokens>` of the generated source (part of the source, so part of the cache key),
    // then the tokens of the environment variable PCL_JIT_OPTS (experiments; jit_cache_key hashes it)
    std::vector<std::string> extra;
    auto add_tokens = [&](const std::string &line) {
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) ++i;
            size_t j = i;
            while (j < line.size() && line[j] != ' ' && line[j] != '\t') ++j;
            if (j > i) extra.push_back(line.substr(i, j - i));
            i = j;
        }
    };
    for (size_t pos = 0; source.compare(pos, 16, "// pcl-jit-opt: ") == 0;) {
        const size_t eol = source.find('\n', pos);
        add_tokens(source.substr(pos + 16, (eol == std::string::npos ? source.size() : eol) - pos - 16));
        if (eol == std::string::npos) break;
        pos = eol + 1;
    }
    if (const char *e = getenv("PCL_JIT_OPTS")) add_tokens(e);
    std::vector<const char *> opts(kJitOpts, kJitOpts + sizeof kJitOpts / sizeof kJitOpts[0]);
    for (const std::string &o : extra) opts.push_back(o.c_str());
    if (g_rtc.CompileProgram(prog, (int)opts.size(), opts.data()) != 0) {
        size_t ls = 0;
        g_jit_note = std::string("hiprtcCompileProgram failed for ") + what;
        if (g_rtc.GetProgramLogSize && g_rtc.GetProgramLog && g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
            std::string log(ls, '\0');
            g_rtc.GetProgramLog(prog, &log[0]);
            g_jit_note += ": " + log.substr(0, 400);
        }
        g_rtc.DestroyProgram(&prog);
        return false;
    }
    const char *lowered = nullptr;
    size_t cs = 0;
    if (!plain_name) g_rtc.GetLoweredName(prog, name_expr, &lowered);
    g_rtc.GetCodeSize(prog, &cs);
    code.resize(cs);
    g_rtc.GetCode(prog, code.data());
    lname = plain_name ? std::string(name_expr) : (lowered ? lowered : "");
    g_rtc.DestroyProgram(&prog);
    return !lname.empty() && cs > 0;
}

// Compile (once per process, device and key) `source` against the kernel headers next to the library and return the kernel
// `name_expr` names (a template instance such as "pcl_hess_kernel_v2<2, 4, 24, true>", or an extern "C" kernel of the source).
hipFunction_t jit_compile(int device, const std::string &key_, const std::string &source, const char *name_expr, bool plain_name) {
    const std::string key = std::to_string(device) + "|" + key_;
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit.find(key);
    if (it != g_jit.end()) {
        if (it->second.failed) return nullptr;
        if (!plain_name) return it->second.fn;
        hipFunction_t f = nullptr;  // several kernels of one generated module
        return hipModuleGetFunction(&f, it->second.mod, name_expr) == hipSuccess ? f : nullptr;
    }
    JitKernel &jk = g_jit[key];
    jk.failed = true;
    Dl_info info;
    if (!dladdr((const void *)&pcl_version, &info) || !info.dli_fname) {
        g_jit_note = "dladdr failed";
        return nullptr;
    }
    std::string dir(info.dli_fname);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    const char *names[] = {"pcl_device_common.hpp", "pcl_kernels_fused_v2.hpp", "pcl_kernel_fused_v3.hpp", "pcl_kernels_hessian.hpp",
                           "pcl_kernel_hessian_v3.hpp", "pcl_kernel_hessian_sparse.hpp", "pcl_kernel_eval_sparse.hpp",
                           "pcl_kernel_fused_sparse.hpp", "pcl_kernel_hess_sparse4.hpp", "pcl_kernel_hess_cols.hpp", "pcl_kernel_hess_cols_parts.hpp"};
    constexpr int NH = 11;
    std::string hdr[NH];
    const char *hdrp[NH];
    for (int i = 0; i < NH; ++i) {
        if (!slurp(dir + "/" + names[i], hdr[i])) {
            g_jit_note = "kernel header not found next to the library: " + dir + "/" + names[i];
            return nullptr;
        }
        hdrp[i] = hdr[i].c_str();
    }
    std::vector<char> code;
    std::string lname = plain_name ? std::string(name_expr) : std::string();
    const std::string ckey = jit_cache_key(source, hdr, NH, plain_name ? "" : name_expr);
    bool from_cache = false;
    std::string cache_file;  // the user-cache file the module came from (removed if it turns out not to load)
    const std::string ucd = cache_enabled() ? user_cache_dir() : std::string();
    if (cache_enabled()) {
        std::vector<char> blob;
        if (read_file(dir + "/prebuilt/" + ckey + ".hsaco", blob))
            from_cache = unpack_module(blob, code, lname, plain_name);
        if (!from_cache && user_cache_trusted(ucd, ckey + ".hsaco") && read_file(ucd + "/" + ckey + ".hsaco", blob)) {
            cache_file = ucd + "/" + ckey + ".hsaco";
            from_cache = unpack_module(blob, code, lname, plain_name);
            if (!from_cache) {  // truncated / corrupt / older format: away with it
                (void)remove(cache_file.c_str());
                cache_file.clear();
            }
        }
    }
    auto load = [&]() {
        jk.mod = nullptr;
        jk.fn = nullptr;
        if (lname.empty() || hipModuleLoadData(&jk.mod, code.data()) != hipSuccess) return false;
        if (hipModuleGetFunction(&jk.fn, jk.mod, lname.c_str()) != hipSuccess) {
            (void)hipModuleUnload(jk.mod);
            jk.mod = nullptr;
            return false;
        }
        return true;
    };
    bool loaded = from_cache && load();
    if (from_cache && !loaded) {  // a cached code object the driver does not take (other ROCm build, damaged file): compile instead of failing for good
        if (!cache_file.empty()) (void)remove(cache_file.c_str());
        from_cache = false;
        code.clear();
        if (!plain_name) lname.clear();
    }
    if (!loaded) {
        if (!rtc_load()) return nullptr;
        if (!rtc_compile(source, hdrp, names, NH, name_expr, plain_name, key_, code, lname)) return nullptr;
        if (!ucd.empty()) {
            mkdir_p(ucd);
            if (user_cache_trusted(ucd)) (void)write_file_atomic(ucd, ckey + ".hsaco", pack_module(code, lname));
        }
        if (!load()) {
            g_jit_note = "hipModuleLoadData / hipModuleGetFunction failed";
            return nullptr;
        }
    }
    jk.failed = false;
    if (from_cache)
        ++g_jit_cache_hits;
    else
        ++g_jit_compiles;
    return jk.fn;
}
hipFunction_t jit_function(int device, const char *instance) {
    static const std::string src = "#include \"pcl_device_common.hpp\"\n#include \"pcl_kernels_fused_v2.hpp\"\n#include \"pcl_kernel_fused_v3.hpp\"\n"
                                   "#include \"pcl_kernels_hessian.hpp\"\n#include \"pcl_kernel_hessian_v3.hpp\"\n";
    return jit_compile(device, instance, src, instance, false);
}
// Source of the pattern-compiled kernels of one system (pcl_codegen.hpp)
std::string sparse_source(const pcl_codegen::SpPlan &plan) {
    return "#include \"pcl_device_common.hpp\"\n" + pcl_codegen::apply_functions(plan) + "#include \"pcl_kernel_hessian_sparse.hpp\"\n#include \"pcl_kernel_eval_sparse.hpp\"\n";
}
// Source of the pattern-compiled fused residual + Jacobian kernel of one system at Pade order 2q (pcl_codegen_v4.hpp)
// (np: tiles of the powers of G -- v4_power_tiles)
std::string v4_source(const pcl_codegen::V4Plan &plan, int q, int np, int variant = 0, int tickets = 0) {
    // tickets: 0 the static work splits | 1 with the slice-ticket roles | 2 the resident evaluator (static split, the evaluation as a device function)
    return std::string("#include \"pcl_device_common.hpp\"\n#define SP4_TICKETS ") + (tickets == 1 ? "1\n" : "0\n") + (tickets == 2 ? "#define SP4_RESIDENT 1\n" : "") + pcl_codegen::v4_functions(plan, q, np, variant) + "#include \"pcl_kernel_fused_sparse.hpp\"\n";
}

// The general-order Hessian kernels live at the register limit (256 per lane: a chain column, two output vectors, the product's accumulators).  With the
// scheduler's default pressure estimate the allocator spills and, worse, rotates whole register sets to open a slot (137 v_mov_b64 per pass of
// pcl_hess_cols_kernel at config 3); with the GCN pressure trackers the same source compiles without scratch and with 21 moves per pass
// (lab/probes/hc_isa/hc_isa_stats.py).  The fused kernels are not register-bound and compile to the same code either way.
static const char kHessJitOpt[] = "// pcl-jit-opt: -mllvm -amdgpu-use-amdgpu-trackers=1\n";

// ... and of the Hessian-of-the-Lagrangian kernel of the same family (pcl_kernel_hess_sparse4.hpp; any order)
std::string v4_hess_source(const pcl_codegen::V4Plan &plan, int q, int variant = 0, int split = 1) {  // variant (profile builds): SH_VARIANT of the kernel (bits >= 16), bit 8: the gather-dot reads nine columns at a time
    return std::string(kHessJitOpt) + "#include \"pcl_device_common.hpp\"\n#define SH_VARIANT " + std::to_string(variant) + "\n#define SH_SPLIT " + std::to_string(split) + "\n" + pcl_codegen::v4_functions(plan, q, 1, variant & 8, true) + "#include \"pcl_kernel_hess_sparse4.hpp\"\n";
}
// ... one wave per group of state columns (pcl_kernel_hess_cols.hpp; any order)
std::string v4_hess_cols_source(const pcl_codegen::V4Plan &plan, int q, int variant = 0) {
    return std::string(kHessJitOpt) + "#include \"pcl_device_common.hpp\"\n" + pcl_codegen::v4_functions(plan, q, 1, (variant & 7) | 8, true) + "#include \"pcl_kernel_hess_cols.hpp\"\n";
}
static size_t hess_cols_rtile_doubles(int d, int m) {  // HC_RTS of the kernel: a column group's R tile, in LDS and in memory
    const int cpw = 32 / (m + 1);
    return ((size_t)cpw * (2 * d + 1) + 1) & ~(size_t)1;
}
// HC_RT_OFF + n_tiles HC_RTS of the kernel (n_tiles < 0: HC_NR, the wave forms the chain of the R_a itself; 1: R-chain waves leave them in memory)
static size_t hess_cols_lds_bytes(int d, int m, int q, int gt_total, int n_tiles = -1) {
    const int cpw = 32 / (m + 1);
    const size_t rt_off = ((size_t)((m + 1) * cpw + 2 * cpw) * (2 * d + 1) + 16 + ((size_t)m * 2 * (((size_t)gt_total + 2) / 3) + 1) / 2 + 1 + 1) & ~(size_t)1;  // (+ 1: the state word)
    return (rt_off + (size_t)(n_tiles < 0 ? (q > 2 ? q - 2 : 0) : n_tiles) * hess_cols_rtile_doubles(d, m)) * sizeof(double);
}
}  // namespace

// Compile `source` (a generated module) with hiprtc and leave the code object in `out_dir` under its content hash: no device needed.
static int prebuild_source(const std::string &source, const char *name_expr, const char *out_dir, std::string &err) {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    Dl_info info;
    if (!dladdr((const void *)&pcl_version, &info) || !info.dli_fname) {
        err = "dladdr failed";
        return PCL_EHIP;
    }
    std::string dir(info.dli_fname);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    const char *names[] = {"pcl_device_common.hpp", "pcl_kernels_fused_v2.hpp", "pcl_kernel_fused_v3.hpp", "pcl_kernels_hessian.hpp",
                           "pcl_kernel_hessian_v3.hpp", "pcl_kernel_hessian_sparse.hpp", "pcl_kernel_eval_sparse.hpp",
                           "pcl_kernel_fused_sparse.hpp", "pcl_kernel_hess_sparse4.hpp", "pcl_kernel_hess_cols.hpp", "pcl_kernel_hess_cols_parts.hpp"};
    constexpr int NH = 11;
    std::string hdr[NH];
    const char *hdrp[NH];
    for (int i = 0; i < NH; ++i) {
        if (!slurp(dir + "/" + names[i], hdr[i])) {
            err = "kernel header not found next to the library: " + dir + "/" + names[i];
            return PCL_EHIP;
        }
        hdrp[i] = hdr[i].c_str();
    }
    const std::string ckey = jit_cache_key(source, hdr, NH, "");
    const std::string odir = out_dir && out_dir[0] ? std::string(out_dir) : dir + "/prebuilt";
    std::vector<char> blob;
    {
        std::vector<char> code0;
        std::string name0;
        if (read_file(odir + "/" + ckey + ".hsaco", blob) && unpack_module(blob, code0, name0, false)) return PCL_OK;  // already there, and whole
    }
    if (!rtc_load()) {
        err = g_jit_note;
        return PCL_EHIP;
    }
    std::vector<char> code;
    std::string lname;
    if (!rtc_compile(source, hdrp, names, NH, name_expr, true, "prebuild", code, lname)) {
        err = g_jit_note;
        return PCL_EHIP;
    }
    if (!write_file_atomic(odir, ckey + ".hsaco", pack_module(code, lname))) {
        err = "cannot write " + odir + "/" + ckey + ".hsaco";
        return PCL_EHIP;
    }
    return PCL_OK;
}
// the pattern-compiled modules of one system, as a context of that system would compile them on first use (no device needed):
//   what 0  fused residual + Jacobian + residual-only kernels at order 2q | 1  general-order Hessian, one workgroup per interval |
//        2  ... two workgroups per interval | 3  the order-4 Hessian / value-table module (q ignored) | 4  the fused module with the slice-ticket
//        roles | 5  general-order Hessian, one wave per group of state columns | 6  the fused module of the resident evaluator
extern "C" int pcl_jit_prebuild(int d, int m, const double *G0, int n_g0, const double *Gj, int q, int what, const char *out_dir) {
    if (d < 1 || d > 32 || m < 0 || m > 6 || n_g0 < 1 || q < 1 || q > 5 || what < 0 || what > 6 || !G0 || (m > 0 && !Gj)) return fail(nullptr, PCL_EINVAL, "pcl_jit_prebuild: bad argument");
#ifndef PCL_LAB
    if (what == 6) return fail(nullptr, PCL_ENOTIMPL, "pcl_jit_prebuild: the resident evaluator's module belongs to lab builds (-DPCL_LAB)");
#endif
    std::string src, err;
    const char *kernel = "pcl_fused_sparse_kernel";
    if (what == 3) {
        src = sparse_source(pcl_codegen::make_plan(d, m, G0, n_g0, Gj));
        kernel = "pcl_hess_sparse_kernel";
    } else {
        const pcl_codegen::V4Plan plan = pcl_codegen::make_v4_plan(d, m, G0, n_g0, Gj);
        if (!plan.ok) return fail(nullptr, PCL_ESHAPE, "pcl_jit_prebuild: the pattern-compiled kernels do not take this system");
        const int np = v4_power_tiles(d, m, q, 160 * 1024);
        if (!np) return fail(nullptr, PCL_ESHAPE, "pcl_jit_prebuild: tiles exceed LDS");
        src = what == 0 ? v4_source(plan, q, np) : what == 4 ? v4_source(plan, q, np, 0, 1) : what == 6 ? v4_source(plan, q, np, 0, 2) : what == 5 ? v4_hess_cols_source(plan, q) : v4_hess_source(plan, q, 0, what);
        if (what == 6)
            kernel = "pcl_fused_sparse_resident";
        else if (what && what != 4)
            kernel = what == 5 ? "pcl_hess_cols_kernel" : "pcl_hess_sparse4_kernel";
    }
    const int rc = prebuild_source(src, kernel, out_dir, err);
    return rc == PCL_OK ? rc : fail(nullptr, rc, "pcl_jit_prebuild: %s", err.c_str());
}

// Inspection hooks of the fused pattern-compiled kernel (no device needed): its generated source, and the generator's term
// tables applied on the host to one column (y = G(u) x; n_g0 drifts span the union pattern, the first one is applied).
extern "C" int pcl_codegen_source_v4(int d, int m, const double *G0, int n_g0, const double *Gj, int q, int what, char *buf, int64_t cap, int64_t *needed) {
    if (d < 1 || d > 32 || m < 0 || m > 6 || n_g0 < 1 || q < 1 || q > 5 || (what != 0 && what != 1 && what != 5) || !G0 || (m > 0 && !Gj) || !needed) return PCL_EINVAL;
    const pcl_codegen::V4Plan plan = pcl_codegen::make_v4_plan(d, m, G0, n_g0, Gj);
    if (!plan.ok) return PCL_ESHAPE;
    const int np = v4_power_tiles(d, m, q, 160 * 1024);
    if (!np) return PCL_ESHAPE;
    const std::string src = what == 5 ? v4_hess_cols_source(plan, q) : what == 1 ? v4_hess_source(plan, q) : v4_source(plan, q, np);
    *needed = (int64_t)src.size() + 1;
    if (buf && cap > 0) {
        const size_t nb = std::min<size_t>((size_t)cap - 1, src.size());
        memcpy(buf, src.data(), nb);
        buf[nb] = '\0';
    }
    return PCL_OK;
}
#ifdef PCL_LAB  // include/piccolo_hip_lab.h (tests/test_abi_cpu.py checks the term tables through a host-only shim of the same two functions)
extern "C" int pcl_codegen_apply_v4(int d, int m, const double *G0, int n_g0, const double *Gj, const double *u, const double *x, double *y, int transposed) {
    if (d < 1 || d > 32 || m < 0 || m > 6 || n_g0 < 1 || !G0 || (m > 0 && (!Gj || !u)) || !x || !y) return PCL_EINVAL;
    const pcl_codegen::V4Plan plan = pcl_codegen::make_v4_plan(d, m, G0, n_g0, Gj);
    if (!plan.ok) return PCL_ESHAPE;
    if (transposed)
        pcl_codegen::v4_reference_apply_t(plan, G0, u, x, y);
    else
        pcl_codegen::v4_reference_apply(plan, G0, Gj, u, x, y);
    return PCL_OK;
}
#endif

// Inspection hook: the generated source of the pattern-compiled kernels for a system (needs no device).
extern "C" int pcl_codegen_source(int d, int m, const double *G0, const double *Gj, char *buf, int64_t cap, int64_t *needed) {
    if (d < 1 || d > 32 || m < 0 || m > 6 || !G0 || (m > 0 && !Gj) || !needed) return PCL_EINVAL;
    const pcl_codegen::SpPlan plan = pcl_codegen::make_plan(d, m, G0, 1, Gj);
    const std::string src = sparse_source(plan);
    *needed = (int64_t)src.size() + 1;
    if (buf && cap > 0) {
        const size_t nb = std::min<size_t>((size_t)cap - 1, src.size());
        memcpy(buf, src.data(), nb);
        buf[nb] = '\0';
    }
    return PCL_OK;
}
