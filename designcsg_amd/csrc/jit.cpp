// jit.cpp -- builds the per-scene gfx950 code object.
//
// Replaces Evaluator::build (Evaluator.cpp:45-112: clBuildProgram of k2.cl + scene.cl with
// "-D STACK_MEMORY_PER_PIXEL=64 -w", Utils.cpp:78-81).  The user's brush bodies are OpenCL C text
// inside the scene format, so that text alone goes through the OpenCL C *language front-end*
// (no OpenCL runtime) to gfx950 bitcode; it is linked with the hand-written HIP kernels
// (kernels.hip built to bitcode with DCSG_EXTERN_BANK) and lowered to one code object:
//   comgr: COMPILE_SOURCE_WITH_DEVICE_LIBS_TO_BC -> LINK_BC_TO_BC -> CODEGEN_BC_TO_RELOCATABLE
//          -> LINK_RELOCATABLE_TO_EXECUTABLE
// all in-process (libamd_comgr), so nothing is exec'ed from a process that owns the GPU.
// Code objects are cached on disk keyed by a hash of every input.
#include "jit.hpp"

#include <amd_comgr/amd_comgr.h>  // types and enums only: the library itself is dlopen'ed, see comgr()
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <mutex>
#include <sstream>
#include <vector>

#include "scene.hpp"

namespace dcsg {

#include "embedded_cl.inc"  // kScenePrelude / kSceneEpilogue, generated from the .cl files by build.py

static const char* kIsa = "amdgcn-amd-amdhsa--gfx950";

// libamd_comgr is loaded by absolute path from the ROCm installation whose compiler produced
// the kernel bitcode (dcsg_kernels_*.bc), with RTLD_LOCAL | RTLD_DEEPBIND.  A PyTorch process already holds a
// bundled libamd_comgr.so.3 (older LLVM, same SONAME) that cannot read this bitcode, so linking
// against the SONAME would silently pick the wrong one.
struct ComgrApi {
    void* handle = nullptr;
    std::string error;
#define DCSG_COMGR_FN(name) decltype(&::name) name = nullptr;
    DCSG_COMGR_FN(amd_comgr_create_data)
    DCSG_COMGR_FN(amd_comgr_release_data)
    DCSG_COMGR_FN(amd_comgr_set_data)
    DCSG_COMGR_FN(amd_comgr_set_data_name)
    DCSG_COMGR_FN(amd_comgr_get_data)
    DCSG_COMGR_FN(amd_comgr_create_data_set)
    DCSG_COMGR_FN(amd_comgr_destroy_data_set)
    DCSG_COMGR_FN(amd_comgr_data_set_add)
    DCSG_COMGR_FN(amd_comgr_action_data_count)
    DCSG_COMGR_FN(amd_comgr_action_data_get_data)
    DCSG_COMGR_FN(amd_comgr_create_action_info)
    DCSG_COMGR_FN(amd_comgr_destroy_action_info)
    DCSG_COMGR_FN(amd_comgr_action_info_set_isa_name)
    DCSG_COMGR_FN(amd_comgr_action_info_set_language)
    DCSG_COMGR_FN(amd_comgr_action_info_set_option_list)
    DCSG_COMGR_FN(amd_comgr_action_info_set_logging)
    DCSG_COMGR_FN(amd_comgr_do_action)
    DCSG_COMGR_FN(amd_comgr_status_string)
#undef DCSG_COMGR_FN
};

// Which compiler builds the code objects: the file libamd_comgr would be loaded from and its size (a ROCm upgrade changes
// at least one of them; the time stamp is left out: the same image unpacked elsewhere must hit the cache), so that cached code objects of another compiler are never reused.  Only
// stat(2): a cache hit must not pay for loading the compiler.
static std::vector<std::string> comgr_candidates()
{
    std::vector<std::string> c;
    if (const char* e = getenv("DCSG_COMGR_PATH")) c.push_back(e);
    const char* rocm = getenv("ROCM_PATH");
    c.push_back(std::string(rocm ? rocm : "/opt/rocm") + "/lib/libamd_comgr.so.3");
    c.push_back("/opt/rocm/lib/libamd_comgr.so.3");
    c.push_back("libamd_comgr.so.3");
    return c;
}
static std::string comgr_identity(const std::vector<std::string>& candidates)
{
    for (auto& c : candidates) {
        struct stat st;
        if (stat(c.c_str(), &st) == 0)
            return c + ":" + std::to_string((long long)st.st_size);
    }
    return "unknown";
}

// Loaded once per process (the first caller's candidate list decides), under a lock: contexts on different host threads
// -- dcsg_export_multi, a background build next to another context's load -- may get here together.
static ComgrApi& comgr(const std::vector<std::string>* candidates = nullptr)
{
    static ComgrApi api;
    static std::mutex once;
    std::lock_guard<std::mutex> lock(once);
    if (api.handle || !api.error.empty() || !candidates) return api;
    for (auto& c : *candidates) {
#if defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)
        api.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);  // the sanitizer runtimes refuse RTLD_DEEPBIND (tools/sanitize_host.sh)
#else
        api.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
#endif
        if (api.handle) break;
        api.error += std::string(dlerror()) + "\n";
    }
    if (!api.handle) return api;
    api.error.clear();
#define DCSG_COMGR_FN(name)                                           \
    api.name = (decltype(api.name))dlsym(api.handle, #name);          \
    if (!api.name) api.error += "libamd_comgr lacks " #name "\n";
    DCSG_COMGR_FN(amd_comgr_create_data)
    DCSG_COMGR_FN(amd_comgr_release_data)
    DCSG_COMGR_FN(amd_comgr_set_data)
    DCSG_COMGR_FN(amd_comgr_set_data_name)
    DCSG_COMGR_FN(amd_comgr_get_data)
    DCSG_COMGR_FN(amd_comgr_create_data_set)
    DCSG_COMGR_FN(amd_comgr_destroy_data_set)
    DCSG_COMGR_FN(amd_comgr_data_set_add)
    DCSG_COMGR_FN(amd_comgr_action_data_count)
    DCSG_COMGR_FN(amd_comgr_action_data_get_data)
    DCSG_COMGR_FN(amd_comgr_create_action_info)
    DCSG_COMGR_FN(amd_comgr_destroy_action_info)
    DCSG_COMGR_FN(amd_comgr_action_info_set_isa_name)
    DCSG_COMGR_FN(amd_comgr_action_info_set_language)
    DCSG_COMGR_FN(amd_comgr_action_info_set_option_list)
    DCSG_COMGR_FN(amd_comgr_action_info_set_logging)
    DCSG_COMGR_FN(amd_comgr_do_action)
    DCSG_COMGR_FN(amd_comgr_status_string)
#undef DCSG_COMGR_FN
    return api;
}

std::string library_dir()
{
    Dl_info info;
    if (dladdr((void*)&library_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

bool read_file(const std::string& path, std::string& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

static bool write_file_atomic(const std::string& path, const std::string& data)
{
    std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return false;
        f.write(data.data(), (std::streamsize)data.size());
        if (!f) return false;
    }
    return rename(tmp.c_str(), path.c_str()) == 0;
}

// A directory is trusted for cached code objects only if it belongs to this user and nobody else can write to it:
// whatever lies there is handed to hipModuleLoadData.
static bool private_dir(const std::string& d)
{
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && access(d.c_str(), W_OK) == 0;
}

// DCSG_JIT_CACHE, else <library dir>/jitcache (the in-tree build), else $XDG_CACHE_HOME/dcsg_jitcache or
// ~/.cache/dcsg_jitcache (mode 0700).  Empty string: no usable directory, the cache is off.
static std::string cache_dir()
{
    std::vector<std::string> candidates;
    if (const char* env = getenv("DCSG_JIT_CACHE")) {
        candidates.push_back(env);
    } else {
        candidates.push_back(library_dir() + "/jitcache");
        const char* xdg = getenv("XDG_CACHE_HOME");
        const char* home = getenv("HOME");
        if (xdg && *xdg) candidates.push_back(std::string(xdg) + "/dcsg_jitcache");
        else if (home && *home) {
            mkdir((std::string(home) + "/.cache").c_str(), 0700);
            candidates.push_back(std::string(home) + "/.cache/dcsg_jitcache");
        }
    }
    for (auto& d : candidates) {
        mkdir(d.c_str(), 0700);
        if (private_dir(d)) return d;
    }
    return "";
}

// a cached file is loaded only if it is a regular file of this user and looks like a 64-bit little-endian AMDGPU ELF
static bool read_cached_code_object(const std::string& path, std::string& out)
{
    struct stat st;
    if (lstat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != getuid()) return false;
    if (!read_file(path, out) || out.size() < 64) return false;
    const unsigned char* e = (const unsigned char*)out.data();
    const unsigned machine = e[18] | (e[19] << 8);
    return e[0] == 0x7f && e[1] == 'E' && e[2] == 'L' && e[3] == 'F' && e[4] == 2 && e[5] == 1 && machine == 224;  // EM_AMDGPU
}

struct ComgrSet {
    amd_comgr_data_set_t s{};
    bool ok = false;
    ComgrSet() { ok = comgr().amd_comgr_create_data_set(&s) == AMD_COMGR_STATUS_SUCCESS; }
    ~ComgrSet()
    {
        if (ok) comgr().amd_comgr_destroy_data_set(s);
    }
};

static bool add_data(amd_comgr_data_set_t set, amd_comgr_data_kind_t kind, const char* name, const std::string& bytes)
{
    amd_comgr_data_t d;
    if (comgr().amd_comgr_create_data(kind, &d) != AMD_COMGR_STATUS_SUCCESS) return false;
    bool ok = comgr().amd_comgr_set_data(d, bytes.size(), bytes.data()) == AMD_COMGR_STATUS_SUCCESS &&
              comgr().amd_comgr_set_data_name(d, name) == AMD_COMGR_STATUS_SUCCESS &&
              comgr().amd_comgr_data_set_add(set, d) == AMD_COMGR_STATUS_SUCCESS;
    comgr().amd_comgr_release_data(d);
    return ok;
}

static std::string collect(amd_comgr_data_set_t set, amd_comgr_data_kind_t kind)
{
    std::string out;
    size_t n = 0;
    if (comgr().amd_comgr_action_data_count(set, kind, &n) != AMD_COMGR_STATUS_SUCCESS) return out;
    for (size_t i = 0; i < n; i++) {
        amd_comgr_data_t d;
        if (comgr().amd_comgr_action_data_get_data(set, kind, i, &d) != AMD_COMGR_STATUS_SUCCESS) continue;
        size_t sz = 0;
        if (comgr().amd_comgr_get_data(d, &sz, nullptr) == AMD_COMGR_STATUS_SUCCESS && sz) {
            std::string b(sz, '\0');
            comgr().amd_comgr_get_data(d, &sz, &b[0]);
            out += b;
        }
        comgr().amd_comgr_release_data(d);
    }
    return out;
}

static bool run_action_(amd_comgr_action_kind_t kind, amd_comgr_language_t lang, const std::vector<const char*>& opts,
                        amd_comgr_data_set_t in, amd_comgr_data_set_t out, std::string& log);

// DCSG_JIT_PROFILE=1: seconds per comgr action on stderr (where a cold build spends its time); set per call of
// jit_scene_module from its JitEnv
static thread_local bool t_profile = false;
static bool run_action(amd_comgr_action_kind_t kind, amd_comgr_language_t lang, const std::vector<const char*>& opts,
                       amd_comgr_data_set_t in, amd_comgr_data_set_t out, std::string& log)
{
    if (!t_profile) return run_action_(kind, lang, opts, in, out, log);
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    bool ok = run_action_(kind, lang, opts, in, out, log);
    clock_gettime(CLOCK_MONOTONIC, &b);
    fprintf(stderr, "dcsg jit: comgr action %d: %.3f s\n", (int)kind, (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec));
    return ok;
}

static bool run_action_(amd_comgr_action_kind_t kind, amd_comgr_language_t lang, const std::vector<const char*>& opts,
                        amd_comgr_data_set_t in, amd_comgr_data_set_t out, std::string& log)
{
    // One comgr action at a time per PROCESS: two host threads compiling at once (two contexts, or a background build
    // next to another context's load) write the same globals inside libamd_comgr (ThreadSanitizer, tools/sanitize_host.sh
    // --thread: a std::string assignment to a library global from both).  Compiles are rare and long; serialising them
    // costs a second context nothing it would not also lose to the compiler's own locks.
    static std::mutex comgr_actions;
    std::lock_guard<std::mutex> lock(comgr_actions);
    amd_comgr_action_info_t info;
    if (comgr().amd_comgr_create_action_info(&info) != AMD_COMGR_STATUS_SUCCESS) {
        log += "comgr: create_action_info failed\n";
        return false;
    }
    bool ok = comgr().amd_comgr_action_info_set_isa_name(info, kIsa) == AMD_COMGR_STATUS_SUCCESS;
    if (ok && lang != AMD_COMGR_LANGUAGE_NONE)
        ok = comgr().amd_comgr_action_info_set_language(info, lang) == AMD_COMGR_STATUS_SUCCESS;
    if (ok && !opts.empty())
        ok = comgr().amd_comgr_action_info_set_option_list(info, const_cast<const char**>(opts.data()), opts.size()) ==
             AMD_COMGR_STATUS_SUCCESS;
    if (ok) ok = comgr().amd_comgr_action_info_set_logging(info, true) == AMD_COMGR_STATUS_SUCCESS;
    amd_comgr_status_t st = AMD_COMGR_STATUS_ERROR;
    if (ok) st = comgr().amd_comgr_do_action(kind, info, in, out);
    comgr().amd_comgr_destroy_action_info(info);
    std::string l = collect(out, AMD_COMGR_DATA_KIND_LOG);
    if (st != AMD_COMGR_STATUS_SUCCESS) {
        const char* name = nullptr;
        comgr().amd_comgr_status_string(st, &name);
        log += l;
        log += std::string("comgr action ") + std::to_string((int)kind) + " failed: " + (name ? name : "?") + "\n";
        return false;
    }
    return true;
}

JitEnv JitEnv::from_environment()
{
    JitEnv e;
    if (const char* o = getenv("DCSG_JIT_EXTRA_OPTS")) e.extra_opts = o;
    e.cache_dir = getenv("DCSG_JIT_NO_CACHE") ? std::string() : dcsg::cache_dir();
    e.profile = getenv("DCSG_JIT_PROFILE") != nullptr;
    e.comgr_paths = comgr_candidates();
    return e;
}

bool jit_scene_module(const std::string& scene_cl, const std::string& extra_cl, std::string& hsaco, std::string& log,
                      bool* from_cache, bool relaxed, bool spec_set, int part, bool cache_only, const JitEnv* env)
{
    if (from_cache) *from_cache = false;
    std::string kernels_bc;
    std::string bc_path = library_dir() + (relaxed ? "/dcsg_kernels_relaxed_" : "/dcsg_kernels_") + (spec_set ? "spec" : "interp") +
                          (part == JIT_PART_REST ? "_rest.bc" : "_export.bc");
    if (!read_file(bc_path, kernels_bc)) {
        log = "cannot read " + bc_path + " (run __graft_entry__.build())";
        return false;
    }
    std::string source = std::string(kScenePrelude) + scene_cl + kSceneEpilogue + extra_cl, source_key;

    // Extra compiler options for experiments (';'-separated), e.g. DCSG_JIT_EXTRA_OPTS="-mllvm;-vectorize-slp=false".
    // They take part in the cache key.
    const JitEnv env_now = env ? JitEnv() : JitEnv::from_environment();
    const JitEnv& E = env ? *env : env_now;
    t_profile = E.profile;
    std::vector<std::string> extra_opts;
    if (!E.extra_opts.empty()) {
        std::string t = E.extra_opts;
        size_t i = 0;
        while (i <= t.size()) {
            size_t j = t.find(';', i);
            std::string o = t.substr(i, j == std::string::npos ? std::string::npos : j - i);
            if (!o.empty()) extra_opts.push_back(o);
            if (j == std::string::npos) break;
            i = j + 1;
        }
        for (auto& o : extra_opts) source += "\n// opt: " + o;
    }
    // SLP vectorisation is off: on gfx950 the packed-f32 VALU ops it forms (v_pk_mul_f32 / v_pk_add_f32) issue
    // slower than the scalar pairs they replace (measured: sphere sweep +15 %, Design2 +24 % with it off).
    // Exact module: no contraction, IEEE divide/sqrt -- the numerics contract of kernels.hip.  "-w" as in the
    // reference's build options (Utils.cpp:78-81).
    std::vector<const char*> cl_opts, cg_opts;
    if (relaxed) {
        cl_opts = {"-DDCSG_RELAXED=1", "-ffp-contract=fast", "-cl-mad-enable", "-w", "-O3", "-fno-slp-vectorize"};
        cg_opts = {"-O3", "-ffp-contract=fast", "-fno-slp-vectorize"};
    } else {
        cl_opts = {"-ffp-contract=off", "-cl-fp32-correctly-rounded-divide-sqrt", "-w", "-O3", "-fno-slp-vectorize"};
        cg_opts = {"-O3", "-ffp-contract=off", "-fno-slp-vectorize"};
    }
    for (auto& o : extra_opts) {
        cl_opts.push_back(o.c_str());
        cg_opts.push_back(o.c_str());
    }
    // everything the code object depends on goes into the key: the text, the kernels' bitcode, the ISA, every
    // compiler option, and the identity of the compiler (libamd_comgr's path and size stand in for its version)
    {
        std::string ident = std::string("\n// isa: ") + kIsa;
        for (auto o : cl_opts) ident += std::string("\n// cl: ") + o;
        for (auto o : cg_opts) ident += std::string("\n// cg: ") + o;
        ident += "\n// comgr: " + comgr_identity(E.comgr_paths);
        source_key = source + ident;
    }
    uint64_t h1 = fnv1a64(source_key.data(), source_key.size(), 0xcbf29ce484222325ull);
    h1 = fnv1a64(kernels_bc.data(), kernels_bc.size(), h1);
    uint64_t h2 = fnv1a64(source_key.data(), source_key.size(), 0x84222325cbf29ce4ull);
    h2 = fnv1a64(kernels_bc.data(), kernels_bc.size(), h2);
    char key[40];
    snprintf(key, sizeof(key), "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    const std::string& cdir = E.cache_dir;
    const std::string cpath = cdir.empty() ? std::string() : cdir + "/" + key + ".hsaco";
    if (!cpath.empty() && read_cached_code_object(cpath, hsaco)) {
        if (from_cache) *from_cache = true;
        return true;
    }
    hsaco.clear();
    if (cache_only) return false;

    ComgrApi& api = comgr(&E.comgr_paths);
    if (!api.handle || !api.error.empty()) {
        log = "cannot load the ROCm code object manager (libamd_comgr.so.3):\n" + api.error;
        return false;
    }
    // 1. brush text (OpenCL C) -> gfx950 bitcode
    ComgrSet src, bc;
    if (!src.ok || !bc.ok || !add_data(src.s, AMD_COMGR_DATA_KIND_SOURCE, "scene.cl", source)) {
        log = "comgr: cannot create source data";
        return false;
    }
    if (!run_action(AMD_COMGR_ACTION_COMPILE_SOURCE_WITH_DEVICE_LIBS_TO_BC, AMD_COMGR_LANGUAGE_OPENCL_2_0, cl_opts, src.s,
                    bc.s, log))
        return false;
    // 2. + the HIP kernels' bitcode
    if (!add_data(bc.s, AMD_COMGR_DATA_KIND_BC, "dcsg_kernels.bc", kernels_bc)) {
        log = "comgr: cannot add kernel bitcode";
        return false;
    }
    ComgrSet linked, reloc, exe;
    if (!linked.ok || !reloc.ok || !exe.ok) {
        log = "comgr: cannot create data sets";
        return false;
    }
    if (!run_action(AMD_COMGR_ACTION_LINK_BC_TO_BC, AMD_COMGR_LANGUAGE_NONE, {}, bc.s, linked.s, log)) return false;
    // 3. optimise across the seam (sdf_bank inlines into the interpreter) and emit ISA
    if (!run_action(AMD_COMGR_ACTION_CODEGEN_BC_TO_RELOCATABLE, AMD_COMGR_LANGUAGE_NONE, cg_opts, linked.s, reloc.s, log))
        return false;
    if (!run_action(AMD_COMGR_ACTION_LINK_RELOCATABLE_TO_EXECUTABLE, AMD_COMGR_LANGUAGE_NONE, {}, reloc.s, exe.s, log))
        return false;
    hsaco = collect(exe.s, AMD_COMGR_DATA_KIND_EXECUTABLE);
    if (hsaco.size() < 64) {
        log += "comgr produced no executable\n";
        return false;
    }
    if (!cpath.empty()) write_file_atomic(cpath, hsaco);  // best effort
    return true;
}

}  // namespace dcsg

// Exposed for build.py / tests: compile one scene.cl into the cache without touching a GPU.
extern "C" int dcsg_jit_prebuild(const char* scene_cl, char* log_out, size_t log_cap, int* from_cache)
{
    std::string hsaco, log;
    dcsg::Scene none;
    bool ok = true, cached = true;
    for (int part = 0; part < 2 && ok; part++) {
        bool c1 = false;
        ok = dcsg::jit_scene_module(scene_cl ? scene_cl : "", dcsg::specialised_source(none, false, false, false), hsaco, log, &c1,
                                    false, false, part);
        cached = cached && c1;
    }
    if (from_cache) *from_cache = cached ? 1 : 0;
    if (log_out && log_cap) {
        strncpy(log_out, log.c_str(), log_cap - 1);
        log_out[log_cap - 1] = '\0';
    }
    return ok ? (int)0 : -4;
}
