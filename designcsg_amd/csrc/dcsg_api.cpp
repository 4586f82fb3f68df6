// dcsg_api.cpp -- C ABI of libdcsg_hip.so (include/dcsg.h): host side of the MI355X export path.
//
// Replaces Evaluator.{h,cpp} (OpenCL dispatch of k2), ISV.hpp (lazy block cache), the host octree
// mesher / projection loop of cms/main/Headers/mesh.hpp, the writers of utils.hpp and the driver
// MyFrame::OnExportInner (DesignCSG.cpp:638-790).  There is no CPU fallback: without a HIP device
// every entry point fails with DCSG_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dcsg.h"
#include "dcsg_device_params.h"
#include "jit.hpp"
#include "scene.hpp"

using namespace dcsg;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Variants of the evaluation kernels (kernels.hip): [0] generic interpreter (register program, stack in
// LDS, any program), [1] / [2] stack-program interpreter with a register-resident stack of depth <= 4 / <= 8,
// [3] per-scene specialised code (scene modules only), [4] stack interpreter with tile culling (stand-alone
// module only: needs the known Lipschitz bounds of the built-in brushes; eval_points[4] aliases [1]).
struct Kernels {
    hipFunction_t sweep[5] = {}, eval_points[5] = {}, bbox[5] = {}, project[5] = {};
    hipFunction_t project_points = nullptr;  // per-scene code: a lane per stencil point (kernels.hip, dcsg_project_points_body)
    hipFunction_t classify = nullptr, complex_edges = nullptr, scan = nullptr, emit = nullptr, set_ad = nullptr, test_sqrt = nullptr, test_normal = nullptr;
    hipFunction_t set_ad_rest = nullptr;  // the rest part of a scene module has its own `arbitrary_data` pointer
    hipFunction_t sweep_d[5] = {}, sweep_f[5] = {};  // sweep[v] with the outputs fixed at compile time: SDF only / SDF + normals
    hipFunction_t sweep_nodes[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // node-wise sweep per variant
    hipFunction_t dd_insert = nullptr, dd_resolve = nullptr, dd_count = nullptr, dd_gather = nullptr, dd_scatter = nullptr;  // vertex dedup
    hipFunction_t rt_mark = nullptr, rt_count = nullptr, rt_emit = nullptr;  // opt-in retopologize
    hipFunction_t cells_mark = nullptr, cells_collect = nullptr, cells_build = nullptr, project_cells = nullptr;  // cell lists of many-object scenes (built-in module)
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

thread_local std::string g_create_error;

}  // namespace

struct dcsg_ctx {
    int device = 0;
    int n_cu = 256;
    size_t max_lds = 160 * 1024;  // LDS one workgroup may request (hipDeviceProp_t::maxSharedMemoryPerMultiProcessor)
    hipStream_t stream = nullptr;
    hipModule_t builtin_mod = nullptr, scene_mod = nullptr;
    // A scene module comes in two code objects (jit.hpp): scene_mod = the export part, loaded by dcsg_load_scene;
    // scene_mod_rest = dense sweeps, point evaluation, T-junction repair, test kernels -- loaded on first use
    // (need_rest).  After a cold build of the export part the rest is compiled by a background host thread (comgr only,
    // no HIP call) while the caller already exports.
    hipModule_t scene_mod_rest = nullptr;
    // The background build OWNS what it reads and writes (its inputs are copies, its outputs its own members): nothing of
    // the context is touched from the thread, so an early return of the call that started it -- or a later call that
    // rewrites the context's rest_* fields -- cannot race with it.  The context reads `hsaco / log / ok` only after join().
    struct RestJob {
        std::string cl, extra;
        bool relaxed = false, spec = false;
        JitEnv env;                 // the JIT's environment switches as the starting call saw them (no getenv in the thread)
        std::string hsaco, log;
        bool ok = false;
        std::thread thread;
        ~RestJob()
        {
            if (thread.joinable()) thread.join();
        }
    };
    std::unique_ptr<RestJob> rest_job;
    bool rest_wanted = false;   // the loaded scene has a rest part that is not loaded yet
    int rest_kind = 0;
    std::string rest_cl, rest_extra;   // what the rest part of the LOADED scene is built from (host thread only)
    bool rest_relaxed = false, rest_spec = false;
    DevBuf d_node_mask;                 // node-wise sweep: which nodes of the sweep level hold lattice values
    hipModule_t preview_mod = nullptr;  // scene.cl as given + kernels, built on the first dcsg_render_preview of a scene
    hipFunction_t preview_fn = nullptr;
    std::string scene_cl_text;
    Kernels kb, ks;       // built-in bank / scene module
    Kernels* k = nullptr;  // active
    bool scene_loaded = false, jit = false, specialised = false;
    bool light_scene = false;  // at most four evaluations of standard brushes: its sweeps are store-bound (sweep_grid)
    DcsgLattice nz_lattice;    // sweep_device: the lattice last checked for a -0.0 coordinate, and the answer
    bool nz_valid = false, nz_result = false;
    int numerics = 0;  // DCSG_NUMERICS_EXACT / _RELAXED for the scenes loaded from now on (dcsg_set_numerics)
    bool scene_relaxed = false;
    int variant = 0;  // index into Kernels::sweep[] etc., chosen from the scene's register count
    int n_object_evals = 0;  // brush evaluations per evaluation of the scene (leaf steps of the stack program)
    Scene scene;
    std::vector<float> ad_host;
    std::vector<uint8_t> tri_table;
    DevBuf d_blob, d_ad, d_tri_table, d_pts, d_out, d_nodes[2], d_info, d_block_sums, d_small, d_tris, d_lat_sdf,
        d_lat_nrm, d_dd_table, d_dd_rep, d_dd_slot, d_dd_compact, d_rt_keep, d_rt_out;
    // cell lists (kernels.hip "Cell lists"): built on the first projection of a many-object scene, kept until the next scene
    int pattern_key = -1;      // the node sweep's point pattern in d_pattern: shift | step << 8 | edges << 24
    unsigned n_pattern = 0;
    DevBuf d_pending, d_pattern, d_cells, d_cell_entries, d_cell_want, d_cell_words, d_cell_work;
    DcsgCellGrid cell_grid = {};
    bool cells_valid = false;
    double cells_built_ms = 0;  // host-side time spent building cells of the loaded scene so far (launch to completion)
    uint64_t cell_entries = 0, cells_built = 0;
    // stage A's by-product for the Z-slab cuts: probes with |sdf| < one probe spacing per probe plane z = half_neg + cd (k - 128)
    uint32_t probe_z_hist[256] = {};
    float probe_cd = 0, probe_half_neg = 0;
    uint64_t n_tris = 0;
    uint64_t last_unique_vertices = 0;  // distinct positions projected by the last dcsg_project_triangles
    // pinned host words for the small read-backs (counts, totals, the bounding-box words): a copy into pageable memory goes
    // through the runtime's staging path and costs tens of microseconds more, a dozen times per export
    unsigned char* h_words = nullptr;  // 32 KiB
    void* h_pinned = nullptr;  // host staging of the WHOLE triangle buffer (opt-in writers only, which need all of it at once)
    size_t h_pinned_cap = 0;
    // The reference writers stream: a ring of pinned slots that downloads fill and the two writers drain, allocated once
    // per context (pinning memory costs 5-25 ms per 100 MB on the pool's boxes, and the whole-buffer staging of rounds 1-3
    // paid it on every first export of a larger mesh: 6.7 ms of a 25.9 ms Design2 1024^3 export)
    static constexpr int kRingSlots = 4;
    static constexpr size_t kRingSlotBytes = (size_t)16 << 20;
    unsigned char* h_ring = nullptr;
    size_t ring_slot_bytes = 0;
    hipEvent_t ring_ev[kRingSlots] = {};
    // Pinning the ring's 64 MiB takes 10-12 ms (0.18 ms per MiB on the pool's boxes): a host thread started by dcsg_create
    // does it while the context is built and the scene loads, instead of the first export (Design2 1024^3, scene files ->
    // output files in a fresh process: 21 -> 12 ms).  The thread owns `job` until it is joined (take_ring).
    struct RingJob {
        std::thread th;
        int device = 0;
        unsigned char* p = nullptr;
        size_t slot_bytes = 0;
    };
    std::unique_ptr<RingJob> ring_job;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t sweep_ev[8][2] = {};  // timing events of the dense sweeps, used in turn (sweep_device)
    int sweep_ev_next = 0;
    hipStream_t copy_stream = nullptr;          // downloads that overlap the projection of the next chunk
    hipEvent_t chunk_ready[8] = {}, chunk_done[8] = {};
    bool have_sweep_time = false;
    bool time_device_sweeps = false;  // dcsg_sweep_lattice_device records timing events (dcsg_set_sweep_timing)
    std::string err;
    std::string profile_create, profile_load;  // host-side timing breakdown of dcsg_create / the last dcsg_load_scene
};

namespace {

// "name=ms;" entries appended to a profile string
struct Lap {
    std::string& out;
    double t;
    explicit Lap(std::string& o) : out(o), t(now_s()) {}
    void mark(const char* name)
    {
        double n = now_s();
        char buf[96];
        snprintf(buf, sizeof(buf), "%s=%.3f;", name, (n - t) * 1e3);
        out += buf;
        t = n;
    }
};

int fail(dcsg_ctx* c, int code, const std::string& msg)
{
    if (c)
        c->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, DCSG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

int ensure(dcsg_ctx* c, DevBuf& b, size_t bytes, bool preserve = false, size_t keep_bytes = 0)
{
    if (bytes <= b.cap && b.p) return 0;
    size_t want = bytes < 256 ? 256 : bytes;
    if (preserve && b.cap) want = want > b.cap + b.cap / 2 ? want : b.cap + b.cap / 2;
    // small buffers grow sixteen-fold (at most to 64 MiB): the mesher's per-level buffers quadruple from level to level, and
    // every growth is a hipFree (0.17 ms) on a first export's critical path
    const size_t roomy = (size_t)64 << 20;
    if (b.cap && want < roomy) want = std::max(want, std::min(b.cap * 16, roomy));
    void* np = nullptr;
    HIPCHK(c, hipMalloc(&np, want));
    if (b.p) {
        if (preserve && keep_bytes) {
            hipError_t e = hipMemcpyAsync(np, b.p, keep_bytes, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) {
                hipFree(np);
                return fail(c, DCSG_ERR_HIP, std::string("device realloc copy: ") + hipGetErrorString(e));
            }
        }
        HIPCHK(c, hipFree(b.p));
    }
    b.p = np;
    b.cap = want;
    return 0;
}

void release(DevBuf& b)
{
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// kind: 0 the stand-alone module (interpreters + culling kernels), 1 a scene module of the "spec" set (per-scene
// specialised kernels only), 2 a scene module of the "interp" set (interpreters) -- designcsg_amd/build.py
// part: 0 = both parts from one module (the stand-alone code object), 1 = the export part of a scene module
// (stage A, node-wise sweep, mesher, projection, de-duplication), 2 = its rest part (dense sweeps, point evaluation,
// T-junction repair, test kernels) -- kernels.hip, DCSG_PART_*.  Entries of the other part are left as they are.
int load_kernels(dcsg_ctx* c, hipModule_t mod, Kernels& k, int kind, int part = 0)
{
    static const char* suffix[4] = {"", "_s4", "_s8", "_sp"};
    const bool exp = part != 2, rest = part != 1;
    if (part != 2) k = Kernels();
    for (int v = (kind == 1 ? 3 : 0); v < (kind == 1 ? 4 : 3); v++) {
        struct {
            const char* name;
            hipFunction_t* f;
            bool in_export;
        } tab[] = {{"dcsg_sweep", &k.sweep[v], false}, {"dcsg_eval_points", &k.eval_points[v], false}, {"dcsg_bbox", &k.bbox[v], true},
                   {"dcsg_project", &k.project[v], true}, {"dcsg_sweep_nodes", &k.sweep_nodes[v], true}};
        for (auto& t : tab)
            if (t.in_export ? exp : rest) HIPCHK(c, hipModuleGetFunction(t.f, mod, (std::string(t.name) + suffix[v]).c_str()));
        if (v == 3 && exp) HIPCHK(c, hipModuleGetFunction(&k.project_points, mod, "dcsg_project_sp_points"));
        if (rest) {
            HIPCHK(c, hipModuleGetFunction(&k.sweep_d[v], mod, (std::string("dcsg_sweep") + suffix[v] + "_d").c_str()));
            HIPCHK(c, hipModuleGetFunction(&k.sweep_f[v], mod, (std::string("dcsg_sweep") + suffix[v] + "_f").c_str()));
        }
    }
    struct {
        const char* name;
        hipFunction_t* f;
        bool in_export;
    } tab[] = {{"dcsg_mesh_classify", &k.classify, true}, {"dcsg_mesh_complex_edges", &k.complex_edges, true},
               {"dcsg_scan_blocks", &k.scan, true}, {"dcsg_mesh_emit", &k.emit, true},
               {"dcsg_test_sqrt", &k.test_sqrt, false},       {"dcsg_dedup_insert", &k.dd_insert, true},
               {"dcsg_test_normal", &k.test_normal, false},
               {"dcsg_dedup_count", &k.dd_count, true},      {"dcsg_dedup_gather", &k.dd_gather, true},
               {"dcsg_dedup_scatter", &k.dd_scatter, true},  {"dcsg_dedup_resolve", &k.dd_resolve, true},
               {"dcsg_retopo_mark", &k.rt_mark, false},
               {"dcsg_retopo_count", &k.rt_count, false},     {"dcsg_retopo_emit", &k.rt_emit, false}};
    for (auto& t : tab)
        if (t.in_export ? exp : rest) HIPCHK(c, hipModuleGetFunction(t.f, mod, t.name));
    if (kind == 0) {
        HIPCHK(c, hipModuleGetFunction(&k.sweep[4], mod, "dcsg_sweep_c4"));
        HIPCHK(c, hipModuleGetFunction(&k.sweep_d[4], mod, "dcsg_sweep_c4_d"));
        HIPCHK(c, hipModuleGetFunction(&k.sweep_f[4], mod, "dcsg_sweep_c4_f"));
        HIPCHK(c, hipModuleGetFunction(&k.sweep_nodes[4], mod, "dcsg_sweep_nodes_c4"));
        HIPCHK(c, hipModuleGetFunction(&k.bbox[4], mod, "dcsg_bbox_c4"));
        k.project[4] = nullptr;  // culling scenes project through the cell lists (project_cells)
        HIPCHK(c, hipModuleGetFunction(&k.cells_mark, mod, "dcsg_cells_mark"));
        HIPCHK(c, hipModuleGetFunction(&k.cells_collect, mod, "dcsg_cells_collect"));
        HIPCHK(c, hipModuleGetFunction(&k.cells_build, mod, "dcsg_cells_build"));
        HIPCHK(c, hipModuleGetFunction(&k.project_cells, mod, "dcsg_project_cells"));
        k.eval_points[4] = k.eval_points[1];  // arbitrary points have no spatial coherence to exploit
    } else {
        HIPCHK(c, hipModuleGetFunction(part == 2 ? &k.set_ad_rest : &k.set_ad, mod, "dcsg_set_arbitrary_data"));
    }
    return 0;
}

int launch(dcsg_ctx* c, hipFunction_t f, unsigned grid, unsigned block, size_t lds, hipStream_t s, void** args)
{
    if (grid == 0) return 0;
    HIPCHK(c, hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, (unsigned)lds, s, args, nullptr));
    return 0;
}

// ISV3D64 geometry for box{center,diameters}, res (ISV.hpp:91-109); float ops exactly as written there.
DcsgLattice make_lattice(const float center[3], const float diam[3], int res)
{
    DcsgLattice L;
    for (int k = 0; k < 3; k++) {
        volatile float h = 0.5f * diam[k];  // v3f_scale(bb.diameters, 0.5)
        L.base[k] = center[k] - h;          // v3f_sub(bb.center, ...)
        L.diam[k] = diam[k];
        L.center[k] = center[k];
        L.half[k] = diam[k] / 2.0f;
    }
    L.w = (float)res;
    L.inv_w = 1.0f / (float)res;
    L.res = res;
    L.w_pow2 = (res > 0 && (res & (res - 1)) == 0) ? 1 : 0;
    // n / (res+1) by multiply-high, round-up method: exact for every 32-bit n
    uint32_t d = (uint32_t)res + 1u, lg = 0;
    while ((1ull << lg) < d) lg++;
    L.n1_magic = (unsigned)((((1ull << lg) - d) << 32) / d + 1);
    L.n1_shift = lg ? lg - 1 : 0;
    return L;
}

unsigned sweep_grid(const dcsg_ctx* c, unsigned tiles, unsigned tiles_per_block = 0)
{
    // 8 resident 256-thread blocks per CU, x8 for tail balance (measured 4..256 per CU on the 513^3 and 1025^3 sweeps with
    // settled clocks: 64 is the fastest for the fused kernel at both sizes; one tile per block is 1.5-2.5x slower).
    // tiles_per_block > 0 (the SDF-only dense sweep of a scene of at most four standard-brush evaluations; a heavy brush
    // such as Design2's is arithmetic-bound and 2 % slower this way): as many blocks as give each that many tiles, between 64 and 512
    // per CU.  A block's tiles are `grid` tiles apart, so a launch has as many write fronts moving through the lattice
    // as a block has tiles; the store-bound SDF-only kernel is fastest with 8 of them at either size (1025^3: 0.89 ms
    // with 64 blocks per CU = 64 tiles per block, 0.83 / 0.80 / 0.75 ms with 192 / 256 / 512; 513^3, where 64 per CU
    // already means 8 tiles: 0.110 ms, against 0.139 with 32 per CU and 0.124 with 128), the fused kernel is not.
    static const int per_cu = getenv("DCSG_BLOCKS_PER_CU") ? atoi(getenv("DCSG_BLOCKS_PER_CU")) : 0;
    unsigned cap = (unsigned)c->n_cu * (unsigned)(per_cu > 0 ? per_cu : 64);
    if (per_cu <= 0 && tiles_per_block > 0) {
        unsigned want = tiles / tiles_per_block, hi = (unsigned)c->n_cu * 512u;
        cap = want < cap ? cap : (want > hi ? hi : want);
    }
    unsigned grid = tiles < cap ? tiles : cap;
    // whole rounds of resident blocks (8 per CU): 131 841 tiles / 8 = 16 480 blocks would leave 96 blocks to a ninth,
    // nearly empty round that ends a block's duration (a tenth of a 0.1 ms sweep) after everything else; rounded down, the
    // grid-stride loop gives the first blocks one more tile instead
    const unsigned slots = (unsigned)c->n_cu * 8u;
    const bool whole_rounds = !getenv("DCSG_NO_WHOLE_ROUNDS");
    if (whole_rounds && grid > slots) grid -= grid % slots;
    return grid;
}

// How a scene's code object is obtained: the stand-alone module with the HIP-native brush bank, or a
// module compiled from its scene.cl (+ the generated specialisation of its build procedure).
struct ModulePlan {
    bool needs_jit = false, specialise = false, null0 = true, relaxed = false, light = false;
    std::string cl, extra;
};

bool plan_scene_module(const Scene& s, const char* scene_cl, ModulePlan& p, std::string& err, bool relaxed = false)
{
    p.cl = scene_cl ? scene_cl : "";
    p.relaxed = relaxed;
    bool builtin_only;  // every brush in use is one of sd0..sd4 with its standard body
    if (!scene_cl) {
        if (s.max_brush_id > 4) {
            err = "scene uses brush id " + std::to_string(s.max_brush_id) + " but no scene.cl was given";
            return false;
        }
        builtin_only = true;
    } else {
        builtin_only = s.max_brush_id <= 4 && scene_cl_is_builtin_only(p.cl);
    }
    if (s.lds_bytes() > 160 * 1024) {
        err = "scene tables need " + std::to_string(s.lds_bytes()) + " B of LDS (> 160 KiB)";
        return false;
    }
    // Small stack-disciplined programs are specialised: the whole evaluation becomes straight-line code.
    int n_evals = 0;
    for (int k = 0; k < s.n_stack_prog; k++) n_evals += (s.stack_prog[4 * k] <= 35 || (s.stack_prog[4 * k] >= 41 && s.stack_prog[4 * k] <= 43)) ? 1 : 0;
    const char* env_spec = getenv("DCSG_SPECIALISE");  // max object evaluations to specialise; 0 disables
    int spec_limit = env_spec ? atoi(env_spec) : 48;
    p.specialise = s.n_stack_prog > 0 && n_evals <= spec_limit && !getenv("DCSG_FORCE_VARIANT");  // A/B runs of the interpreters
    // Larger programs of the design's OWN brushes (arbitrary OpenCL C is the norm: scenecompiler.py:227-241) are specialised
    // too, in the THREADED form (csrc/specialise.cpp: emit_threaded_prelude): the program as a chain of calls of per-brush
    // functions with the objects' constants in a table -- a few instructions per object, so the compile time stays that of
    // a small scene, and nothing is decoded at run time.  Scenes of the standard brushes alone take the tile-culling kernels
    // instead (they skip most of the objects; this would evaluate all of them).
    const char* env_thr = getenv("DCSG_SPECIALISE_THREADED");  // max object evaluations for the threaded form; 0 disables
    const int threaded_limit = env_thr ? atoi(env_thr) : 1024;
    bool threaded = false;
    if (!p.specialise && s.n_stack_prog > 0 && !builtin_only && n_evals <= threaded_limit && spec_limit > 0 && !getenv("DCSG_FORCE_VARIANT")) {
        p.specialise = true;
        threaded = true;
    }
    p.needs_jit = !(builtin_only && !p.specialise) || relaxed;  // the relaxed arithmetic exists as a scene module only
    p.light = builtin_only && n_evals <= 4;
    if (p.needs_jit) {
        if (builtin_only) p.cl = builtin_scene_cl();
        p.null0 = builtin_only || scene_cl_null_brush_is_const(p.cl);
        p.extra = specialised_source(s, p.null0, p.specialise, builtin_only && !getenv("DCSG_NO_SIMPLIFY"), threaded);
    }
    return true;
}

int need_scene(dcsg_ctx* c)
{
    if (!c) return DCSG_ERR_INVALID;
    if (!c->scene_loaded) return fail(c, DCSG_ERR_NO_SCENE, "no scene loaded (\"Cannot Export Yet.\")");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, DCSG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return 0;
}

// Starts the background build of a scene's rest part.  The caller has joined the previous job (join_rest_thread).
void start_rest_job(dcsg_ctx* c, const ModulePlan& plan)
{
    c->rest_job.reset(new dcsg_ctx::RestJob());
    dcsg_ctx::RestJob* job = c->rest_job.get();
    job->cl = plan.cl;
    job->extra = plan.extra;
    job->relaxed = plan.relaxed;
    job->spec = plan.specialise;
    job->env = JitEnv::from_environment();  // read here, on the caller's thread
    job->thread = std::thread([job] {
        bool from_cache = false;
        job->ok = jit_scene_module(job->cl, job->extra, job->hsaco, job->log, &from_cache, job->relaxed, job->spec, JIT_PART_REST, false,
                                   &job->env);
    });
}

// Ends the background build of the rest part, if one is running (its result stays in c->rest_job).  One comgr build at a
// time per context: every path that starts a build or replaces the job calls this first.
void join_rest_thread(dcsg_ctx* c)
{
    if (c->rest_job && c->rest_job->thread.joinable()) c->rest_job->thread.join();
}

// Makes the rest part of the scene module available: dense sweeps, point evaluation, T-junction repair, test kernels.
// Loaded on first use -- from the JIT cache, from the background build a cold dcsg_load_scene started, or compiled here.
int need_rest(dcsg_ctx* c)
{
    if (!c->jit || !c->rest_wanted) return 0;
    join_rest_thread(c);
    std::string hsaco, log;
    if (c->rest_job && c->rest_job->ok && !c->rest_job->hsaco.empty()) {
        hsaco.swap(c->rest_job->hsaco);   // the background build of THIS scene (dcsg_load_scene replaces the job per scene)
    } else {
        bool cached = false;
        if (!jit_scene_module(c->rest_cl, c->rest_extra, hsaco, log, &cached, c->rest_relaxed, c->rest_spec, JIT_PART_REST))
            return fail(c, DCSG_ERR_BUILD, "Error when building evaluation program:\n" + log);
    }
    c->rest_job.reset();
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipModuleLoadData(&c->scene_mod_rest, hsaco.data()));
    // any failure from here on unloads the module again: rest_wanted stays set and the next call starts over, without a
    // second copy of the code object left behind
    auto drop = [c](int rc) {
        hipModuleUnload(c->scene_mod_rest);
        c->scene_mod_rest = nullptr;
        return rc;
    };
    int r = load_kernels(c, c->scene_mod_rest, c->ks, c->rest_kind, 2);
    if (r) return drop(r);
    void* adp = c->d_ad.p;
    void* args[] = {&adp};
    if ((r = launch(c, c->ks.set_ad_rest, 1, 64, 0, c->stream, args))) return drop(r);
    // other streams may launch this module's kernels next
    if (hipError_t e = hipStreamSynchronize(c->stream)) return drop(fail(c, DCSG_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)));
    c->rest_wanted = false;
    return 0;
}

int eval_points_device(dcsg_ctx* c, const float* d_xyz, size_t n, int mode, float* d_out, hipStream_t s)
{
    if (int rr = need_rest(c)) return rr;
    const size_t chunk = 256u * 256u * 256u;  // MAX_EVAL_POINTS, Evaluator.h:16
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    for (size_t off = 0; off < n; off += chunk) {
        unsigned cnt = (unsigned)((n - off) < chunk ? (n - off) : chunk);
        const float* in = d_xyz + off * 3;
        float* out = d_out + off * (mode == 0 ? 1 : 3);
        unsigned tiles = (cnt + DCSG_BLOCK - 1) / DCSG_BLOCK;
        void* args[] = {(void*)&blob, &nv, (void*)&in, &cnt, &mode, (void*)&out};
        int r = launch(c, c->k->eval_points[c->variant], sweep_grid(c, tiles), DCSG_BLOCK, c->scene.lds_bytes(), s, args);
        if (r) return r;
    }
    return 0;
}

int eval_points_host(dcsg_ctx* c, const float* xyz, size_t n, int mode, float* out)
{
    int r = need_scene(c);
    if (r) return r;
    if (n == 0) return 0;
    if (!xyz || !out) return fail(c, DCSG_ERR_INVALID, "NULL point / output buffer");
    const size_t chunk = 256u * 256u * 256u;
    const size_t per = mode == 0 ? 1 : 3;
    for (size_t off = 0; off < n; off += chunk) {
        size_t cnt = (n - off) < chunk ? (n - off) : chunk;
        if ((r = ensure(c, c->d_pts, cnt * 3 * sizeof(float)))) return r;
        if ((r = ensure(c, c->d_out, cnt * per * sizeof(float)))) return r;
        HIPCHK(c, hipMemcpyAsync(c->d_pts.p, xyz + off * 3, cnt * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        if ((r = eval_points_device(c, (const float*)c->d_pts.p, cnt, mode, (float*)c->d_out.p, c->stream))) return r;
        HIPCHK(c, hipMemcpyAsync(out + off * per, c->d_out.p, cnt * per * sizeof(float), hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int sweep_device(dcsg_ctx* c, const float center[3], const float diam[3], int res, int z0, int z1, float* d_sdf,
                 float* d_nrm, hipStream_t s, bool time_it)
{
    if (res < 1 || z0 < 0 || z1 > res + 1 || z0 > z1) return fail(c, DCSG_ERR_INVALID, "bad lattice range");
    if (!d_sdf && !d_nrm) return 0;
    if (int rr = need_rest(c)) return rr;
    for (int k = 0; k < 3; k++)  // lattice coordinates stay finite (the specialised kernels rely on it)
        if (!(std::fabs(center[k]) <= 1e29f) || !(std::fabs(diam[k]) <= 1e29f))
            return fail(c, DCSG_ERR_INVALID, "lattice box is not finite");
    const uint64_t n1 = (uint64_t)res + 1;
    const uint64_t total = (uint64_t)(z1 - z0) * n1 * n1;
    if (total >= 0xffffff00ull)
        return fail(c, DCSG_ERR_INVALID, "lattice slab has >= 2^32 points; sweep it in smaller Z ranges");
    DcsgLattice L = make_lattice(center, diam, res);
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    // Timing events of the sweep: a ring of pairs, so that a new sweep never re-records an event whose previous record may
    // still be in flight (back-to-back sweeps of a 0.1 ms kernel otherwise run in lock step with the host)
    hipEvent_t t_ev0 = c->ev0, t_ev1 = c->ev1;
    if (time_it) {
        c->sweep_ev_next = (c->sweep_ev_next + 1) % 8;
        t_ev0 = c->sweep_ev[c->sweep_ev_next][0];
        t_ev1 = c->sweep_ev[c->sweep_ev_next][1];
        HIPCHK(c, hipEventRecord(t_ev0, s));
    }
    // launches cover whole planes, < 2^32 points each (the kernels' flat indices are 32-bit, byte offsets 64-bit): the
    // 1025^3 lattice is one launch (rounds 1-2: 2^30 points, i.e. 1022 planes and a 3-plane tail launch)
    const uint64_t plane = n1 * n1;
    uint64_t planes_per = 0xf0000000ull / plane;
    planes_per = planes_per < 1 ? 1 : planes_per;
    // kernels with the outputs fixed at compile time where they exist (SDF only / SDF + normals)
    hipFunction_t fn = c->k->sweep[c->variant];
    if (d_sdf) fn = d_nrm ? c->k->sweep_f[c->variant] : c->k->sweep_d[c->variant];
    // per-axis coordinate tables behind the scene's LDS area (kernels.hip, "Wave-level lattice indexing"); a lattice
    // too large for them (or with rows shorter than a wave) takes the per-lane index path
    size_t lds = (c->scene.lds_bytes() + 15) & ~(size_t)15;
    int axes_off = -1;
    const bool no_axis_tables = getenv("DCSG_NO_AXIS_TABLES") != nullptr;
    if (n1 >= DCSG_WAVE && lds + 3 * n1 * sizeof(float) <= c->max_lds && c->variant != 4 && !no_axis_tables) {
        axes_off = (int)lds;
        lds += 3 * n1 * sizeof(float);
    }
    // DCSG_SWEEP_NO_NEG_ZERO: no coordinate of this lattice is -0.0 (checked here with the kernels' arithmetic,
    // dcsg_lattice_coord, in the same IEEE operations: this file is built with -ffp-contract=off), so the normal stencil's
    // "+ 0.0f" changes nothing and the sweeps of per-scene code skip their per-pass test for it.
    unsigned flags = 0;
    // (diagnostic switches are read per call, not cached: tests flip them inside one process; four lookups are well under a
    // microsecond of the five a launch costs the host)
    const bool no_rows = getenv("DCSG_NO_ROWS") != nullptr, no_multipass = getenv("DCSG_NO_MULTIPASS") != nullptr;
    // read by the SDF + normal sweeps of per-scene code (dcsg_stencil_spec)
    if (c->variant < 4 && !no_rows && d_nrm) {
        // the answer depends on the lattice alone: remembered for the last one (a bench or a slab loop sweeps the same
        // lattice again and again, and 3 (res+1) volatile float operations are microseconds of host time per launch)
        if (!c->nz_valid || memcmp(&c->nz_lattice, &L, sizeof(L)) != 0) {
            bool neg_zero = false;
            for (int k = 0; k < 3 && !neg_zero; k++)
                for (uint64_t i = 0; i < n1 && !neg_zero; i++) {
                    volatile float t = L.diam[k] * (float)(int)i;
                    volatile float u = L.w_pow2 ? t * L.inv_w : t / L.w;
                    volatile float v = L.base[k] + u;
                    float vv = v;
                    uint32_t b;
                    memcpy(&b, &vv, 4);
                    neg_zero = b == 0x80000000u;
                }
            c->nz_lattice = L;
            c->nz_result = neg_zero;
            c->nz_valid = true;
        }
        if (!c->nz_result) flags |= DCSG_SWEEP_NO_NEG_ZERO;
    }
    if (!no_multipass) flags |= DCSG_SWEEP_MULTIPASS;
    // diagnostic: unused LDS per block, to hold fewer blocks on a CU (occupancy experiments, tools/ab_sweep.py)
    static const size_t lds_pad = getenv("DCSG_SWEEP_LDS_PAD") ? (size_t)atol(getenv("DCSG_SWEEP_LDS_PAD")) : 0;
    if (lds + lds_pad <= c->max_lds) lds += lds_pad;
    for (uint64_t p0 = 0; p0 < (uint64_t)(z1 - z0); p0 += planes_per) {
        uint64_t np = (uint64_t)(z1 - z0) - p0 < planes_per ? (uint64_t)(z1 - z0) - p0 : planes_per;
        unsigned first = (unsigned)(p0 * plane);
        unsigned cnt = (unsigned)(np * plane);
        unsigned work = cnt;
        unsigned per_block = 4u * DCSG_BLOCK;  // a block covers 4 runs of 256 points per pass (dcsg_sweep_body)
        if (c->variant == 4) {
            // one brick per 256-thread block iteration (kernels.hip, dcsg_sweep_c4_body): 16 x 16 x 8 points; 8 x 8 x 8 for
            // scenes with smooth joins, where a longer list costs more than building it less often saves.  Measured at
            // 513^3, fused / SDF only, ms (profiles/r04_brick_shapes.txt):   8x8x4 (rounds 1-3)  8x8x8   16x16x4  16x16x8
            //   flat256     4.20 / 3.34    3.23 / 2.29    3.72 / 2.29    3.53 / 2.03
            //   grouped256 12.89 / 10.70   9.42 / 7.07    9.57 / 6.33    8.90 / 5.47
            //   smooth256  17.93 / 12.00  16.02 / 9.61   19.37 / 10.47  19.09 / 9.90
            // DCSG_BRICK=<wide 0|1>,<depth> overrides (measurements).
            int wide = c->scene.has_smooth ? 0 : 1, depth = 8;
            if (const char* e = getenv("DCSG_BRICK")) {
                int w = 1, d = 8;
                if (sscanf(e, "%d,%d", &w, &d) == 2 && d >= 1 && d <= 64 && (w || d % 4 == 0)) wide = w ? 1 : 0, depth = d;
            }
            const uint64_t side = wide ? 16 : 8, bs = (n1 + side - 1) / side;
            work = (unsigned)(bs * bs * ((np + depth - 1) / depth) * DCSG_BLOCK);
            per_block = DCSG_BLOCK;
            axes_off = depth | (wide << 8);
        }
        unsigned tiles = (work + per_block - 1) / per_block;
        void* args[] = {(void*)&blob, &nv, &L, &z0, &first, &cnt, (void*)&d_sdf, (void*)&d_nrm, &axes_off, &flags};
        unsigned grid = sweep_grid(c, tiles, (d_sdf && !d_nrm && c->light_scene) ? 8u : 0u);
        int r = launch(c, fn, grid, DCSG_BLOCK, lds, s, args);
        if (r) return r;
    }
    if (time_it) {
        HIPCHK(c, hipEventRecord(t_ev1, s));
        c->have_sweep_time = true;
    } else {
        c->have_sweep_time = false;  // "the most recent sweep" was not timed: dcsg_last_sweep_ms says so instead of
                                     // returning an older sweep's duration
    }
    return 0;
}

}  // namespace

// ======================================================================================
extern "C" {

const char* dcsg_last_error(const dcsg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int dcsg_create(dcsg_ctx** out, int device_id)
{
    if (!out) return fail(nullptr, DCSG_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DCSG_ERR_NO_DEVICE,
                    std::string("no HIP device visible (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                        "): libdcsg_hip has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(nullptr, DCSG_ERR_INVALID, "device_id out of range");
    dcsg_ctx* c = new dcsg_ctx();
    c->device = device_id;
    if (const char* e = getenv("DCSG_NUMERICS")) c->numerics = strcmp(e, "relaxed") == 0 ? DCSG_NUMERICS_RELAXED : DCSG_NUMERICS_EXACT;
    Lap lap(c->profile_create);
    auto bail = [&](int code) {
        g_create_error = c->err;
        *out = nullptr;
        dcsg_destroy(c);
        return code;
    };
#define CK(expr)                                                                                        \
    do {                                                                                                \
        hipError_t e2 = (expr);                                                                         \
        if (e2 != hipSuccess) {                                                                         \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e2);                                 \
            return bail(DCSG_ERR_HIP);                                                                  \
        }                                                                                               \
    } while (0)
    CK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, device_id));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // dynamic LDS a single workgroup may ask for: the CU's whole LDS on gfx950 (163 840 B); sharedMemPerBlock (64 KiB) is
    // the static-allocation limit and does not apply to hipModuleLaunchKernel's sharedMemBytes
    c->max_lds = prop.maxSharedMemoryPerMultiProcessor >= 65536 ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 65536;
    if (c->max_lds > 160 * 1024) c->max_lds = 160 * 1024;
    lap.mark("hip_runtime_and_device");
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipHostMalloc((void**)&c->h_words, 32768, hipHostMallocDefault));
    CK(hipEventCreate(&c->ev0));
    for (auto& pr : c->sweep_ev) {
        CK(hipEventCreate(&pr[0]));
        CK(hipEventCreate(&pr[1]));
    }
    CK(hipEventCreate(&c->ev1));
    CK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 8; k++) {
        CK(hipEventCreateWithFlags(&c->chunk_ready[k], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&c->chunk_done[k], hipEventDisableTiming));
    }
    if (!getenv("DCSG_NO_RING_PREALLOC")) {
        c->ring_job.reset(new dcsg_ctx::RingJob);
        dcsg_ctx::RingJob* job = c->ring_job.get();
        job->device = device_id;
        job->slot_bytes = dcsg_ctx::kRingSlotBytes;
        job->th = std::thread([job] {
            void* p = nullptr;
            if (hipSetDevice(job->device) == hipSuccess &&
                hipHostMalloc(&p, dcsg_ctx::kRingSlots * job->slot_bytes, hipHostMallocDefault) == hipSuccess)
                job->p = (unsigned char*)p;
        });
    }
    std::string dir = library_dir(), bytes;
    if (!read_file(dir + "/dcsg_builtin.hsaco", bytes)) {
        c->err = "cannot read " + dir + "/dcsg_builtin.hsaco (run __graft_entry__.build())";
        return bail(DCSG_ERR_IO);
    }
    lap.mark("stream_events_read_code_object");
    CK(hipModuleLoadData(&c->builtin_mod, bytes.data()));
    lap.mark("module_load_builtin");
    if (load_kernels(c, c->builtin_mod, c->kb, 0)) return bail(DCSG_ERR_HIP);
    lap.mark("kernel_lookup");
    std::string table_txt;
    if (!read_file(dir + "/../data/lookupTable.txt", table_txt)) {
        c->err = "cannot read packaged lookupTable.txt next to the library";
        return bail(DCSG_ERR_IO);
    }
    *out = c;  // dcsg_set_lookup_table reports through ctx
    int r = dcsg_set_lookup_table(c, table_txt.c_str());
    if (r) {
        *out = nullptr;
        return bail(r);
    }
    c->ad_host.assign(DCSG_ARBITRARY_DATA_POINTS, 0.0f);
    lap.mark("lookup_table");
    // The arbitrary-data table exists from the start (all zeros), uploaded and read back through pageable host memory:
    // the HIP runtime sets up its staging buffers for pageable copies on the first one of a process (about 7 ms, which
    // rounds 1-2 paid in the first dcsg_load_scene); a host creates its context once, at start-up.
    if (ensure(c, c->d_ad, DCSG_ARBITRARY_DATA_POINTS * sizeof(float))) {
        *out = nullptr;
        return bail(DCSG_ERR_HIP);
    }
    CK(hipMemcpy(c->d_ad.p, c->ad_host.data(), DCSG_ARBITRARY_DATA_POINTS * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->ad_host.data(), c->d_ad.p, 4096, hipMemcpyDeviceToHost));
    lap.mark("arbitrary_data_table_and_copy_path");
    return 0;
#undef CK
}

void dcsg_destroy(dcsg_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_blob,  &c->d_ad,         &c->d_tri_table, &c->d_pts,  &c->d_out,     &c->d_nodes[0],
                      &c->d_nodes[1], &c->d_info,    &c->d_block_sums, &c->d_small, &c->d_tris, &c->d_lat_sdf,
                      &c->d_lat_nrm, &c->d_dd_table, &c->d_dd_rep, &c->d_dd_slot, &c->d_dd_compact, &c->d_node_mask,
                      &c->d_rt_keep, &c->d_rt_out, &c->d_pending, &c->d_pattern, &c->d_cells, &c->d_cell_entries, &c->d_cell_want, &c->d_cell_words, &c->d_cell_work};
    for (auto* b : bufs) release(*b);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_words) hipHostFree(c->h_words);
    if (c->ring_job) {
        if (c->ring_job->th.joinable()) c->ring_job->th.join();
        if (c->ring_job->p) hipHostFree(c->ring_job->p);
        c->ring_job.reset();
    }
    if (c->h_ring) hipHostFree(c->h_ring);
    for (auto& e : c->ring_ev)
        if (e) hipEventDestroy(e);
    join_rest_thread(c);
    if (c->scene_mod) hipModuleUnload(c->scene_mod);
    if (c->scene_mod_rest) hipModuleUnload(c->scene_mod_rest);
    if (c->preview_mod) hipModuleUnload(c->preview_mod);
    if (c->builtin_mod) hipModuleUnload(c->builtin_mod);
    if (c->ev0) hipEventDestroy(c->ev0);
    for (auto& pr : c->sweep_ev)
        for (auto& e : pr)
            if (e) hipEventDestroy(e);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (int k = 0; k < 8; k++) {
        if (c->chunk_ready[k]) hipEventDestroy(c->chunk_ready[k]);
        if (c->chunk_done[k]) hipEventDestroy(c->chunk_done[k]);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// Kernel k1 (k1.cl:479-578; DrawPane.cpp:373-470 is its host side): the 640 x 480 preview image of the loaded scene.
// The materials are OpenCL C text of scene.cl, so the image comes from a module built from that text (cached on
// disk like every scene module); a scene loaded without scene.cl gets the standard brushes and materials.
int dcsg_render_preview(dcsg_ctx* c, const float campos[3], const float right[3], const float up[3], const float forward[3],
                        unsigned char* rgb_out)
{
    int r = need_scene(c);
    if (r) return r;
    if (!campos || !right || !up || !forward || !rgb_out) return fail(c, DCSG_ERR_INVALID, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->preview_fn) {
        join_rest_thread(c);  // one comgr action at a time per context: the background build of the rest part first
        std::string text = c->scene_cl_text.empty() ? std::string(builtin_scene_cl()) : c->scene_cl_text;
        std::string hsaco, log;
        Scene none;
        if (!jit_scene_module(text, specialised_source(none, false, false, false), hsaco, log, nullptr, false, false, JIT_PART_REST))
            return fail(c, DCSG_ERR_BUILD, "Error when building preview program:\n" + log);
        HIPCHK(c, hipModuleLoadData(&c->preview_mod, hsaco.data()));
        hipFunction_t set_ad = nullptr;
        HIPCHK(c, hipModuleGetFunction(&c->preview_fn, c->preview_mod, "dcsg_preview"));
        HIPCHK(c, hipModuleGetFunction(&set_ad, c->preview_mod, "dcsg_set_arbitrary_data"));
        void* ad = c->d_ad.p;
        void* args[] = {&ad};
        if ((r = launch(c, set_ad, 1, 64, 0, c->stream, args))) return r;
    }
    const size_t bytes = 640 * 480 * 3;
    if ((r = ensure(c, c->d_out, bytes))) return r;
    DcsgCamera cam;
    for (int k = 0; k < 3; k++) cam.o[k] = campos[k], cam.r[k] = right[k], cam.u[k] = up[k], cam.f[k] = forward[k];
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4(), n_obj = c->scene.n_objects;
    void* out = c->d_out.p;
    void* args[] = {(void*)&blob, &nv, &cam, &n_obj, &out};
    if ((r = launch(c, c->preview_fn, 640 * 480 / DCSG_BLOCK, DCSG_BLOCK, c->scene.lds_bytes(), c->stream, args))) return r;
    HIPCHK(c, hipMemcpyAsync(rgb_out, c->d_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dcsg_set_lookup_table(dcsg_ctx* c, const char* txt)
{
    if (!c || !txt) return fail(c, DCSG_ERR_INVALID, "NULL argument");
    std::vector<uint8_t> t;
    int total = 0;
    std::string err;
    if (!parse_lookup_table(txt, t, total, err)) return fail(c, DCSG_ERR_INVALID, err);
    HIPCHK(c, hipSetDevice(c->device));
    int r = ensure(c, c->d_tri_table, t.size());
    if (r) return r;
    HIPCHK(c, hipMemcpy(c->d_tri_table.p, t.data(), t.size(), hipMemcpyHostToDevice));
    c->tri_table = t;
    return 0;
}

int dcsg_set_arbitrary_data(dcsg_ctx* c, const float* data, size_t items)
{
    if (!c) return DCSG_ERR_INVALID;
    if (items > DCSG_ARBITRARY_DATA_POINTS) return fail(c, DCSG_ERR_INVALID, "more than 131072 arbitrary data items");
    HIPCHK(c, hipSetDevice(c->device));
    if (data && items) memcpy(c->ad_host.data(), data, items * sizeof(float));  // Evaluator.cpp:214
    int r = ensure(c, c->d_ad, DCSG_ARBITRARY_DATA_POINTS * sizeof(float));
    if (r) return r;
    HIPCHK(c, hipMemcpy(c->d_ad.p, c->ad_host.data(), DCSG_ARBITRARY_DATA_POINTS * sizeof(float),
                        hipMemcpyHostToDevice));  // uploads all 131072, Evaluator.cpp:215-223
    if (c->scene_loaded && c->jit) {
        void* ad = c->d_ad.p;
        void* args[] = {&ad};
        if ((r = launch(c, c->ks.set_ad, 1, 64, 0, c->stream, args))) return r;
        if (c->scene_mod_rest && (r = launch(c, c->ks.set_ad_rest, 1, 64, 0, c->stream, args))) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int dcsg_load_scene(dcsg_ctx* c, const char* scene_txt, const char* bp_txt, const char* scene_cl,
                    const float* ad, size_t n_ad)
{
    if (!c) return DCSG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    c->scene_loaded = false;
    c->profile_load.clear();
    Lap lap(c->profile_load);
    Scene s;
    std::string err;
    bool sync_rest = false;
    if (!parse_scene(scene_txt, bp_txt, s, err)) return fail(c, DCSG_ERR_INVALID, err);
    lap.mark("parse_and_optimise");
    ModulePlan plan;
    if (!plan_scene_module(s, scene_cl, plan, err, c->numerics == DCSG_NUMERICS_RELAXED)) return fail(c, DCSG_ERR_INVALID, err);
    lap.mark("plan_and_specialise");
    if (!plan.needs_jit) {
        build_blob(s, true);
        c->k = &c->kb;
        c->jit = false;
        join_rest_thread(c);
        c->rest_job.reset();
        c->rest_wanted = false;
    } else {
        std::string hsaco, log;
        bool cached = false;
        join_rest_thread(c);  // a background build for the previous scene
        if (!jit_scene_module(plan.cl, plan.extra, hsaco, log, &cached, plan.relaxed, plan.specialise, JIT_PART_EXPORT))
            return fail(c, DCSG_ERR_BUILD, "Error when building evaluation program:\n" + log);
        lap.mark(cached ? "code_object_from_cache" : "code_object_compiled");
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->scene_mod) {
            HIPCHK(c, hipModuleUnload(c->scene_mod));
            c->scene_mod = nullptr;
        }
        if (c->scene_mod_rest) {
            HIPCHK(c, hipModuleUnload(c->scene_mod_rest));
            c->scene_mod_rest = nullptr;
        }
        HIPCHK(c, hipModuleLoadData(&c->scene_mod, hsaco.data()));
        lap.mark("module_load_scene");
        int r = load_kernels(c, c->scene_mod, c->ks, plan.specialise ? 1 : 2, 1);
        if (r) return r;
        lap.mark("kernel_lookup");
        // the rest part: from the cache on first use, or -- after a cold build -- compiled by a background thread while
        // the caller goes on (the export needs nothing of it); DCSG_JIT_SYNC=1 builds it here instead
        c->rest_wanted = true;
        c->rest_kind = plan.specialise ? 1 : 2;
        c->rest_cl = plan.cl;
        c->rest_extra = plan.extra;
        c->rest_relaxed = plan.relaxed;
        c->rest_spec = plan.specialise;
        c->rest_job.reset();  // joined above; a result of the previous scene is of no use
        if (!cached) {
            if (getenv("DCSG_JIT_SYNC")) {
                sync_rest = true;
            } else {
                start_rest_job(c, plan);
            }
        }
        build_blob(s, plan.null0);
        c->k = &c->ks;
        c->jit = true;
    }
    c->specialised = c->jit && plan.specialise;
    c->light_scene = plan.light;
    c->scene_relaxed = plan.relaxed;
    // every kernel stages the scene tables in LDS: they must fit one workgroup's share (160 KiB on gfx950)
    if (s.lds_bytes() > c->max_lds)
        return fail(c, DCSG_ERR_INVALID,
                    "scene tables need " + std::to_string(s.lds_bytes()) + " B of LDS (" + std::to_string(s.n_objects) +
                        " objects, " + std::to_string(s.n_steps) + " build steps); the limit is " + std::to_string(c->max_lds) +
                        " B (about 1400 objects at 163840)");
    int r = ensure(c, c->d_blob, s.blob.size() * 4);
    if (r) return r;
    HIPCHK(c, hipMemcpy(c->d_blob.p, s.blob.data(), s.blob.size() * 4, hipMemcpyHostToDevice));
    c->scene = std::move(s);
    const int depth = c->scene.n_stack_prog ? c->scene.stack_depth : 1000;
    c->variant = c->specialised ? 3 : (depth <= 4 ? 1 : (depth <= 8 ? 2 : 0));
    // Tile culling: stand-alone module (built-in brushes), program of top-level items (leaves and one level of
    // groups), enough objects to pay off.
    {
        int n_evals = 0;
        for (int k = 0; k < c->scene.n_stack_prog; k++) n_evals += (c->scene.stack_prog[4 * k] <= 35 || (c->scene.stack_prog[4 * k] >= 41 && c->scene.stack_prog[4 * k] <= 43)) ? 1 : 0;
        c->n_object_evals = n_evals;
        const char* env_cull = getenv("DCSG_CULL");
        int min_objects = env_cull ? atoi(env_cull) : 16;  // 0 disables
        if (!c->jit && c->scene.n_items > 0 && min_objects > 0 && n_evals >= min_objects) c->variant = 4;
    }
    if (const char* force = getenv("DCSG_FORCE_VARIANT")) {  // A/B testing only; never exceeds what the program fits
        int f = atoi(force);
        if (f == 0 || (f == 1 && depth <= 4) || (f == 2 && depth <= 8)) c->variant = f;
        // f == 4: culling is only ever chosen by the rule above
    }
    c->scene_loaded = true;
    c->cells_valid = false;
    c->n_tris = 0;
    c->scene_cl_text = scene_cl ? scene_cl : "";
    if (c->preview_mod) {  // belongs to the previous scene
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipModuleUnload(c->preview_mod));
        c->preview_mod = nullptr;
        c->preview_fn = nullptr;
    }
    std::fill(c->ad_host.begin(), c->ad_host.end(), 0.0f);
    lap.mark("tables_upload");
    // arbitrary_data (DesignCSG.cpp:507-529; all 131 072 floats reach the device, Evaluator.cpp:215-223).  Most designs
    // leave the table empty: an all-zero table is a device-side memset, not a 512 KB copy through a host staging buffer.
    if (ad && n_ad > DCSG_ARBITRARY_DATA_POINTS) return fail(c, DCSG_ERR_INVALID, "more than 131072 arbitrary data items");
    bool all_zero = true;
    if (ad)
        for (size_t i = 0; i < n_ad && all_zero; i++) {
            uint32_t w;
            memcpy(&w, ad + i, 4);
            all_zero = w == 0;
        }
    if ((r = ensure(c, c->d_ad, DCSG_ARBITRARY_DATA_POINTS * sizeof(float)))) return r;
    if (all_zero) {
        HIPCHK(c, hipMemsetAsync(c->d_ad.p, 0, DCSG_ARBITRARY_DATA_POINTS * sizeof(float), c->stream));
    } else {
        memcpy(c->ad_host.data(), ad, n_ad * sizeof(float));
        HIPCHK(c, hipMemcpyAsync(c->d_ad.p, c->ad_host.data(), DCSG_ARBITRARY_DATA_POINTS * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    lap.mark("arbitrary_data_upload");
    if (c->jit) {  // the scene module's `arbitrary_data` pointer; also the first launch on the freshly loaded module
        void* adp = c->d_ad.p;
        void* args[] = {&adp};
        if ((r = launch(c, c->ks.set_ad, 1, 64, 0, c->stream, args))) return r;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lap.mark("first_launch_on_module");
    if (sync_rest) {
        if ((r = need_rest(c))) return r;
        lap.mark("rest_part_compiled_and_loaded");
    }
    return 0;
}

const char* dcsg_load_profile(const dcsg_ctx* c, int which)
{
    if (!c) return "";
    return which == 0 ? c->profile_create.c_str() : c->profile_load.c_str();
}

int dcsg_set_numerics(dcsg_ctx* c, int mode)
{
    if (!c) return DCSG_ERR_INVALID;
    if (mode != DCSG_NUMERICS_EXACT && mode != DCSG_NUMERICS_RELAXED) return fail(c, DCSG_ERR_INVALID, "unknown numerics mode");
    c->numerics = mode;
    return 0;
}

int dcsg_scene_info(const dcsg_ctx* c, int* n_objects, int* n_steps, int* n_slots, int* jit_compiled, int* interpreter,
                    int* stack_depth)
{
    if (!c || !c->scene_loaded) return DCSG_ERR_NO_SCENE;
    if (n_objects) *n_objects = c->scene.n_objects;
    if (n_steps) *n_steps = c->scene.n_steps;
    if (n_slots) *n_slots = c->scene.n_regs;
    if (jit_compiled) *jit_compiled = c->jit ? 1 : 0;
    if (interpreter) *interpreter = c->variant;
    if (stack_depth) *stack_depth = c->scene.n_stack_prog ? c->scene.stack_depth : 0;
    return 0;
}

int dcsg_debug_scene_flags(const dcsg_ctx* c, int* flags)
{
    if (!c || !c->scene_loaded || c->scene.blob.size() < 4) return DCSG_ERR_NO_SCENE;
    if (flags) *flags = (int)c->scene.blob[3];
    return 0;
}

int dcsg_eval_sdf(dcsg_ctx* c, const float* xyz, size_t n, float* out) { return eval_points_host(c, xyz, n, 0, out); }
int dcsg_eval_normal(dcsg_ctx* c, const float* xyz, size_t n, float* out) { return eval_points_host(c, xyz, n, 1, out); }

int dcsg_eval_sdf_device(dcsg_ctx* c, const float* d_xyz, size_t n, float* d_out, void* stream)
{
    int r = need_scene(c);
    return r ? r : eval_points_device(c, d_xyz, n, 0, d_out, (hipStream_t)stream);
}
int dcsg_eval_normal_device(dcsg_ctx* c, const float* d_xyz, size_t n, float* d_out, void* stream)
{
    int r = need_scene(c);
    return r ? r : eval_points_device(c, d_xyz, n, 1, d_out, (hipStream_t)stream);
}

int dcsg_sweep_lattice_device(dcsg_ctx* c, const float center[3], const float diameters[3], int res, int z0, int z1,
                              float* d_sdf, float* d_normals, void* stream)
{
    int r = need_scene(c);
    // timing events only on request (dcsg_set_sweep_timing): two event records per launch put several microseconds between
    // back-to-back sweeps (513^3 SDF-only: 109 -> 10x us per launch over a region of launches)
    return r ? r : sweep_device(c, center, diameters, res, z0, z1, d_sdf, d_normals, (hipStream_t)stream, c->time_device_sweeps);
}

int dcsg_set_sweep_timing(dcsg_ctx* c, int on)
{
    if (!c) return DCSG_ERR_INVALID;
    c->time_device_sweeps = on != 0;
    return 0;
}

int dcsg_last_sweep_ms(dcsg_ctx* c, double* ms)
{
    if (!c || !ms) return DCSG_ERR_INVALID;
    if (!c->have_sweep_time)
        return fail(c, DCSG_ERR_INVALID, "the last sweep was not timed (dcsg_sweep_lattice_device records events only after dcsg_set_sweep_timing(ctx, 1))");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->sweep_ev[c->sweep_ev_next][1]));
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, c->sweep_ev[c->sweep_ev_next][0], c->sweep_ev[c->sweep_ev_next][1]));
    *ms = t;
    return 0;
}

int dcsg_eval_lattice(dcsg_ctx* c, const float center[3], const float diameters[3], int res, int z0, int z1,
                      float* sdf, float* normals)
{
    int r = need_scene(c);
    if (r) return r;
    if (res < 1 || z0 < 0 || z1 > res + 1 || z0 > z1) return fail(c, DCSG_ERR_INVALID, "bad lattice range");
    size_t n = (size_t)(z1 - z0) * (size_t)(res + 1) * (size_t)(res + 1);
    if (n == 0 || (!sdf && !normals)) return 0;
    if (sdf && (r = ensure(c, c->d_lat_sdf, n * sizeof(float)))) return r;
    if (normals && (r = ensure(c, c->d_lat_nrm, n * 3 * sizeof(float)))) return r;
    if ((r = sweep_device(c, center, diameters, res, z0, z1, sdf ? (float*)c->d_lat_sdf.p : nullptr,
                          normals ? (float*)c->d_lat_nrm.p : nullptr, c->stream, true)))
        return r;
    if (sdf) HIPCHK(c, hipMemcpyAsync(sdf, c->d_lat_sdf.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (normals)
        HIPCHK(c, hipMemcpyAsync(normals, c->d_lat_nrm.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Stage A, DesignCSG.cpp:668-712
namespace {
bool slab_cut_planes(const dcsg_ctx* c, const dcsg_export_config& cfg, float center_z, float diameter, int rank, int world, int& z_lo, int& z_hi);
int ensure_coarse_cells(dcsg_ctx* c, hipStream_t s);
}

int dcsg_estimate_bounding_box(dcsg_ctx* c, float bb_half, float center[3], float* diameter, uint64_t* interior)
{
    int r = need_scene(c);
    if (r) return r;
    const int R = 256;
    if (!(std::fabs(bb_half) <= 1e29f)) return fail(c, DCSG_ERR_INVALID, "boundingBoxHalfDiameter is not finite");
    volatile float cd = (float)bb_half / R;  // cube_diameter
    volatile float half_neg = -cd / 2;       // v3f(-cube_diameter / 2)
    double bb_eps = (double)((float)bb_half / R) * 1.0;  // BB_EPSILON
    // DCSG_BBOX_COPIES copies of [0..5] min / max indices, [8..9] interior count, [16..271] probes near the surface per probe
    // plane in z (kernels.hip, dcsg_bbox_commit), reduced here
    const size_t words = (size_t)DCSG_BBOX_WORDS * DCSG_BBOX_COPIES;
    if ((r = ensure(c, c->d_small, std::max<size_t>(4096, words * sizeof(int))))) return r;
    int* seed = (int*)c->h_words;  // pinned; rewritten by the read-back below, which the stream orders behind this upload
    memset(seed, 0, words * sizeof(int));
    for (int k = 0; k < DCSG_BBOX_COPIES; k++)
        for (int j = 0; j < 3; j++) seed[k * DCSG_BBOX_WORDS + j] = 1 << 20, seed[k * DCSG_BBOX_WORDS + 3 + j] = -(1 << 20);
    HIPCHK(c, hipMemcpyAsync(c->d_small.p, seed, words * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    float cdv = cd, hnv = half_neg;
    int* res_ptr = (int*)c->d_small.p;
    {
        void* args[] = {(void*)&blob, &nv, &cdv, &hnv, &bb_eps, (void*)&res_ptr};
        const unsigned per_cu = getenv("DCSG_BBOX_BLOCKS_PER_CU") ? (unsigned)atoi(getenv("DCSG_BBOX_BLOCKS_PER_CU")) : 8u;
        if ((r = launch(c, c->k->bbox[c->variant], (unsigned)c->n_cu * (per_cu ? per_cu : 8u), DCSG_BLOCK, c->scene.lds_bytes(), c->stream, args)))
            return r;
    }
    int* out = (int*)c->h_words;
    HIPCHK(c, hipMemcpyAsync(out, c->d_small.p, words * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t cnt = 0;
    for (int k = 1; k < DCSG_BBOX_COPIES; k++) {  // into copy 0
        const int* o = out + k * DCSG_BBOX_WORDS;
        for (int j = 0; j < 3; j++) out[j] = std::min(out[j], o[j]), out[3 + j] = std::max(out[3 + j], o[3 + j]);
        for (int j = 16; j < DCSG_BBOX_WORDS; j++) out[j] = (int)((uint32_t)out[j] + (uint32_t)o[j]);
    }
    for (int k = 0; k < DCSG_BBOX_COPIES; k++) {
        uint64_t part;
        memcpy(&part, &out[k * DCSG_BBOX_WORDS + 8], 8);
        cnt += part;
    }
    for (int k = 0; k < 256; k++) c->probe_z_hist[k] = (uint32_t)out[16 + k];
    c->probe_cd = cd;
    c->probe_half_neg = half_neg;
    float mn[3] = {0.0f, 0.0f, 0.0f}, mx[3] = {0.0f, 0.0f, 0.0f};  // initialised to 0.0, :688-693
    if (cnt) {
        for (int k = 0; k < 3; k++) {
            volatile float a = cd * (float)out[k];      // v3f_scale(v3f(ix,iy,iz), cube_diameter)
            volatile float b = cd * (float)out[3 + k];
            float pmin = half_neg + a, pmax = half_neg + b;  // v3f_add
            if (pmin < mn[k]) mn[k] = pmin;
            if (pmax > mx[k]) mx[k] = pmax;
        }
    }
    float d[3];
    for (int k = 0; k < 3; k++) {
        volatile float sum = mn[k] + mx[k];
        center[k] = (float)((double)sum * 0.5);  // :703
        d[k] = mx[k] - mn[k];                    // :704
    }
    float inner = d[1] > d[2] ? d[1] : d[2];  // T_max, :706
    *diameter = d[0] > inner ? d[0] : inner;
    if (interior) *interior = cnt;
    return 0;
}

int dcsg_parse_export_config(const char* txt, dcsg_export_config* out)
{
    std::string err;
    if (!out) return DCSG_ERR_INVALID;
    if (!parse_export_config(txt, *out, err)) {
        g_create_error = err;
        return DCSG_ERR_INVALID;
    }
    return 0;
}

}  // extern "C"

namespace {
// Node-wise sweep request of dcsg_export: at level `level` (the first level whose nodes the cull rule can drop for
// good: max(minimumOctreeLevel, 2)) the lattice values inside the nodes that reached the mesher are evaluated into
// d_sdf / d_nrm and the mesher reads them from there on; the levels above sample on demand.
struct NodeSweep {
    int level = 0;
    float* d_sdf = nullptr;
    float* d_nrm = nullptr;  // NULL: normals on demand
    double ms = 0;           // device time of the sweep kernel
    uint64_t nodes = 0, points = 0;
};

int mesh_impl(dcsg_ctx* c, const float center[3], float diameter, const dcsg_export_config* cfg, const float* d_sdf,
              const float* d_normals, int z_lo, int z_hi, int z_own_lo, int z_own_hi, uint64_t* n_triangles,
              dcsg_mesh_stats* stats, void* stream_v, NodeSweep* ns)
{
    int r = need_scene(c);
    if (r) return r;
    if (!cfg || !center) return fail(c, DCSG_ERR_INVALID, "NULL argument");
    if (cfg->grid_level < 1 || cfg->grid_level > 12 || cfg->max_level < 2 || cfg->max_level > cfg->grid_level ||
        cfg->min_level > cfg->max_level || cfg->max_level >= DCSG_MAX_LEVELS)
        return fail(c, DCSG_ERR_INVALID, "export levels must satisfy 2 <= max, min <= max <= grid <= 12");
    if (c->tri_table.empty()) return fail(c, DCSG_ERR_INVALID, "no lookup table");
    hipStream_t s = (hipStream_t)stream_v;
    const int res = 1 << cfg->grid_level;
    if (!d_sdf) z_lo = 1, z_hi = 0;  // nothing resident: every sample is evaluated on demand
    float diam[3] = {diameter, diameter, diameter};

    DcsgMeshParams P;
    memset(&P, 0, sizeof(P));
    P.lat = make_lattice(center, diam, res);
    P.min_level = cfg->min_level;
    P.max_level = cfg->max_level;
    P.grid_level = cfg->grid_level;
    P.q_star = angle_threshold_to_q_star(cfg->complex_surface_threshold);
    P.z_lo = z_lo;
    P.z_hi = z_hi;
    P.has_normals = d_normals ? 1 : 0;
    P.z_own_lo = z_own_lo;
    P.z_own_hi = z_own_hi;
    P.mask_check = getenv("DCSG_MESH_MASK_ALWAYS") ? 1 : 0;
    P.use_spec = c->specialised ? 1 : 0;  // informational: a module of the specialised set samples through its per-scene code

    // mesh.hpp:110-125: levels 0 and 1 are subdivided unconditionally; the 64 level-2 nodes start the
    // traversal.  Child centre = center + ((hd*sign)*0.5f), child hd = 0.5f*hd (octree.hpp:24-32).
    static const float SG[8][3] = {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1},
                                   {-1, 1, 1},  {1, 1, 1},  {1, 1, -1},  {-1, 1, -1}};
    float hd0[3], hd1[3], hd2[3];
    for (int k = 0; k < 3; k++) {
        hd0[k] = diam[k] / 2.0f;  // DesignCSG.cpp:718
        hd1[k] = 0.5f * hd0[k];
        hd2[k] = 0.5f * hd1[k];
    }
    std::vector<float> roots;
    roots.reserve(64 * 3);
    for (int a = 0; a < 8; a++) {
        float c1[3];
        for (int k = 0; k < 3; k++) {
            volatile float t = hd0[k] * SG[a][k];
            volatile float u = 0.5f * t;
            c1[k] = center[k] + u;
        }
        for (int b = 0; b < 8; b++)
            for (int k = 0; k < 3; k++) {
                volatile float t = hd1[k] * SG[b][k];
                volatile float u = 0.5f * t;
                roots.push_back(c1[k] + u);
            }
    }
    unsigned n_nodes = 64;
    if ((r = ensure(c, c->d_nodes[0], (size_t)n_nodes * 3 * sizeof(float)))) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_nodes[0].p, roots.data(), roots.size() * sizeof(float), hipMemcpyHostToDevice, s));
    // d_small: [0..1] totals (u32), level stats at byte 256: [level][4] u64
    if ((r = ensure(c, c->d_small, 4096))) return r;
    HIPCHK(c, hipMemsetAsync(c->d_small.p, 0, 4096, s));
    unsigned* d_totals = (unsigned*)c->d_small.p;
    unsigned long long* d_lstats = (unsigned long long*)((char*)c->d_small.p + 256);

    dcsg_mesh_stats st;
    memset(&st, 0, sizeof(st));
    uint64_t n_tris = 0;
    int cur = 0;
    float hd[3] = {hd2[0], hd2[1], hd2[2]};
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    const unsigned char* d_table = (const unsigned char*)c->d_tri_table.p;
    // on-demand samples of a many-object scene read the cell lists (the coarse cells exist from here on; fine ones where
    // earlier projections of this scene built them)
    DcsgCellLists CL;
    memset(&CL, 0, sizeof(CL));
    if (c->variant == 4 && !getenv("DCSG_MESH_NO_CELLS")) {
        if ((r = ensure_coarse_cells(c, s))) return r;
        CL.G = c->cell_grid;
        CL.cells = c->d_cells.p;
        CL.entries = (const unsigned*)c->d_cell_entries.p;
    }
    // Levels below the minimum level emit no triangle and subdivide every node they keep, and they are small: they run back to
    // back without the host learning their node counts -- the next level is launched for eight nodes per node (an upper
    // bound) and reads its real count from where the scan left it (totals slot level & 1).  The count is read back where it
    // is needed: at the sweep level, or when the bound grows past 2^21 nodes.  (Eight levels cost a 1024^3 export 0.4 ms of
    // round trips before; a rank of a sharded export pays them in full.)
    bool deferred = false;
    const int sync_level = cfg->min_level > 2 ? cfg->min_level : 2;
    for (int level = 2; level <= cfg->max_level && n_nodes > 0; level++) {
        if (deferred && (level >= sync_level || n_nodes > (1u << 21) || getenv("DCSG_MESH_SYNC_LEVELS"))) {
            unsigned* totals = (unsigned*)c->h_words;
            HIPCHK(c, hipMemcpyAsync(totals, d_totals + 2 * ((level - 1) & 1), 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            n_nodes = totals[0];
            deferred = false;
            if (!n_nodes) break;
        }
        const unsigned* n_nodes_dev = deferred ? d_totals + 2 * ((level - 1) & 1) : nullptr;
        P.level = level;
        for (int k = 0; k < 3; k++) P.hd[k] = hd[k];
        if (ns && level == ns->level) {
            // the node list of this level is complete: evaluate the lattice inside these nodes, then read from it
            const int shift = cfg->grid_level - level;
            const size_t N = (size_t)1 << level;
            if ((r = ensure(c, c->d_node_mask, N * N * N))) return r;
            HIPCHK(c, hipMemsetAsync(c->d_node_mask.p, 0, N * N * N, s));
            const float* d_list = (const float*)c->d_nodes[cur].p;
            unsigned char* d_mask = (unsigned char*)c->d_node_mask.p;
            int lvl = level, sh = shift;
            // the slab buffer holds planes z_own_lo .. z_own_hi: nodes of the list that lie outside them (all 64 roots at
            // level 2, children of parents that straddle slabs) are skipped by the kernel
            int zo_lo = z_own_lo, zo_hi = z_own_hi;
            unsigned* d_swept = (unsigned*)((char*)c->d_small.p + 2048);  // zeroed with d_small above
            // culling scenes: one list per eight sibling nodes (kernels.hip, dcsg_sweep_nodes_c4; the other kernels do not
            // read the argument).  Measured, sweep ms with a list per node / per eight: flat256 1024^3 2.0 / 1.1, 512^3 0.6 /
            // 0.4; grouped256 6.0 / 3.3, 1.6 / 1.3; smooth256 10.9 / 9.3 but 3.0 / 3.5 at 512^3 -- with smooth joins the
            // blend radius widens every bound, and a box of eight coarse nodes keeps too much: those scenes group only
            // from level 7 on (nodes of 1/128 of the box).
            // The other kernels take eight siblings per block iteration as well (no list to build: the points of the swept ones are
            // walked as one range, kernels.hip dcsg_node_group).
            int group = (n_nodes % 8 == 0 && (c->variant != 4 || !c->scene.has_smooth || level >= 7) && !getenv("DCSG_NODE_GROUP_1")) ? 8 : 1;
            // The points of a node that the mesher can read (kernels.hip, dcsg_in_sweep_pattern: corners and centres of nodes down
            // to the maximum level, every point along the edges of the levels that make the edge-ambiguity test); the sweep
            // writes those only.  7/9/10: 297 of a node's 729 points.  Nothing to leave out when the maximum level is the
            // grid level (DCSG_SWEEP_ALL_POINTS: the whole node, as rounds 1-3 swept it).
            const int step = 1 << (cfg->grid_level - cfg->max_level);
            const bool edges = cfg->max_level > level;
            unsigned n_pattern = 0;
            const unsigned short* d_pattern = nullptr;
            const int pattern_key = shift | step << 8 | (edges ? 1 << 24 : 0);
            if (step > 1 && !getenv("DCSG_SWEEP_ALL_POINTS") && c->pattern_key == pattern_key) {  // the same export configuration as before
                n_pattern = c->n_pattern;
                d_pattern = (const unsigned short*)c->d_pattern.p;
                P.pattern_step = step;
                P.pattern_edges = edges ? 1 : 0;
            } else if (step > 1 && !getenv("DCSG_SWEEP_ALL_POINTS")) {
                std::vector<unsigned short> pat;
                const int S = 1 << shift;
                for (int lz = 0; lz <= S; lz++)
                    for (int ly = 0; ly <= S; ly++)
                        for (int lx = 0; lx <= S; lx++) {
                            const int m = step - 1, h = step >> 1, e = 2 * step - 1;
                            const bool corner = ((lx | ly | lz) & m) == 0, centre = (lx & m) == h && (ly & m) == h && (lz & m) == h;
                            const bool on_edge = edges && ((lx & e) == 0) + ((ly & e) == 0) + ((lz & e) == 0) >= 2;
                            if (corner || centre || on_edge) pat.push_back((unsigned short)(lx | ly << 5 | lz << 10));
                        }
                n_pattern = (unsigned)pat.size();
                if ((r = ensure(c, c->d_pattern, pat.size() * sizeof(unsigned short)))) return r;
                HIPCHK(c, hipMemcpyAsync(c->d_pattern.p, pat.data(), pat.size() * sizeof(unsigned short), hipMemcpyHostToDevice, s));
                HIPCHK(c, hipStreamSynchronize(s));  // `pat` goes out of scope
                d_pattern = (const unsigned short*)c->d_pattern.p;
                P.pattern_step = step;
                P.pattern_edges = edges ? 1 : 0;
                c->pattern_key = pattern_key;
                c->n_pattern = n_pattern;
            }
            void* args[] = {(void*)&blob, &nv, &P.lat, (void*)&d_list, &n_nodes, &sh, &lvl, (void*)&ns->d_sdf, (void*)&ns->d_nrm,
                            (void*)&d_mask, &P.hd[0], &P.hd[1], &P.hd[2], &zo_lo, &zo_hi, (void*)&d_swept, &group, (void*)&d_pattern, &n_pattern};
            HIPCHK(c, hipEventRecord(c->ev0, s));
            // blocks walk the list with a stride: as many as keep the chip full several times over, each adding its count of
            // swept nodes once (DCSG_NODE_BLOCKS_PER_CU, 0 = a block per node or group as in rounds 1-3)
            const unsigned per_cu = getenv("DCSG_NODE_BLOCKS_PER_CU") ? (unsigned)atoi(getenv("DCSG_NODE_BLOCKS_PER_CU")) : 64u;
            unsigned blocks = (n_nodes + group - 1) / group;
            if (per_cu && blocks > (unsigned)c->n_cu * per_cu) blocks = (unsigned)c->n_cu * per_cu;
            if ((r = launch(c, c->k->sweep_nodes[c->variant], blocks, DCSG_BLOCK, c->scene.lds_bytes(), s, args))) return r;
            HIPCHK(c, hipEventRecord(c->ev1, s));
            unsigned& swept = *(unsigned*)c->h_words;
            HIPCHK(c, hipMemcpyAsync(&swept, d_swept, sizeof(swept), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            ns->ms = ms;
            ns->nodes = swept;  // nodes of this slab that the cull rule kept
            const uint64_t side = ((uint64_t)1 << shift) + 1;
            ns->points = (uint64_t)swept * (n_pattern ? (uint64_t)n_pattern : side * side * side);
            d_sdf = ns->d_sdf;
            d_normals = ns->d_nrm;
            P.z_lo = 0;
            P.z_hi = res;
            P.has_normals = d_normals ? 1 : 0;
            P.node_mask = d_mask;
            P.mask_level = level;
            P.mask_shift = shift;
        }
        unsigned n_blocks = (n_nodes + DCSG_BLOCK - 1) / DCSG_BLOCK;
        if ((r = ensure(c, c->d_info, (size_t)n_nodes * sizeof(unsigned short)))) return r;
        if ((r = ensure(c, c->d_block_sums, (size_t)n_blocks * 5 * sizeof(unsigned)))) return r;  // {children, triangles}, then {culled, subdivided, leaves} per block
        const float* d_nodes = (const float*)c->d_nodes[cur].p;
        unsigned short* d_info = (unsigned short*)c->d_info.p;
        unsigned* d_bs = (unsigned*)c->d_block_sums.p;
        unsigned* d_bstats = d_bs + (size_t)n_blocks * 2;
        unsigned long long* d_ls = d_lstats + 4 * level;
        {
            // pending[0] = nodes waiting for the complex-edge test, then the nodes (kernels.hip, dcsg_mesh_complex_edges)
            if ((r = ensure(c, c->d_pending, ((size_t)n_nodes + 1) * sizeof(unsigned)))) return r;
            unsigned* d_pending = (unsigned*)c->d_pending.p;
            HIPCHK(c, hipMemsetAsync(d_pending, 0, sizeof(unsigned), s));
            const size_t lds = c->scene.lds_bytes();
            void* args[] = {(void*)&blob, &nv, &P, (void*)&d_nodes, &n_nodes, (void*)&d_sdf, (void*)&d_normals,
                            (void*)&d_table, (void*)&d_info, (void*)&d_bs, (void*)&d_bstats, (void*)&d_pending, &CL, (void*)&n_nodes_dev};
            if ((r = launch(c, c->k->classify, n_blocks, DCSG_BLOCK, lds, s, args))) return r;
            if (level >= cfg->min_level && level != cfg->max_level) {  // the levels whose nodes can reach that test
                // a wave per eight pending nodes; their number stays on the device: as many waves as could be needed, at
                // most the chip's fill
                const unsigned blocks = std::min<unsigned>((n_nodes + 31) / 32, (unsigned)c->n_cu * 8u);
                void* cargs[] = {(void*)&blob, &nv, &P, (void*)&d_nodes, (void*)&d_pending, (void*)&d_normals, (void*)&d_table,
                                 (void*)&d_info, (void*)&d_bs, (void*)&d_bstats, &CL};
                if ((r = launch(c, c->k->complex_edges, blocks, DCSG_BLOCK, lds, s, cargs))) return r;
            }
        }
        unsigned* d_level_totals = d_totals + 2 * (level & 1);
        {
            void* args[] = {(void*)&d_bs, &n_blocks, (void*)&d_level_totals, (void*)&d_bstats, (void*)&d_ls};
            if ((r = launch(c, c->k->scan, 1, 1024, 0, s, args))) return r;
        }
        unsigned n_children, n_new_tris = 0;
        const bool defer = level < sync_level && level < cfg->max_level && (uint64_t)n_nodes * 8 <= (1u << 24);
        if (defer) {
            n_children = n_nodes * 8u;  // an upper bound; no triangles below the minimum level
        } else {
            unsigned* totals = (unsigned*)c->h_words;
            HIPCHK(c, hipMemcpyAsync(totals, d_level_totals, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            n_children = totals[0], n_new_tris = totals[1];
        }
        if ((r = ensure(c, c->d_nodes[cur ^ 1], (size_t)(n_children ? n_children : 1) * 3 * sizeof(float)))) return r;
        if ((r = ensure(c, c->d_tris, (size_t)(n_tris + n_new_tris + 1) * 9 * sizeof(float), true,
                        (size_t)n_tris * 9 * sizeof(float))))
            return r;
        {
            float* d_next = (float*)c->d_nodes[cur ^ 1].p;
            float* d_tris = (float*)c->d_tris.p;
            unsigned long long base = n_tris;
            void* args[] = {&P, (void*)&d_nodes, &n_nodes, (void*)&d_info, (void*)&d_bs, (void*)&d_table,
                            (void*)&d_next, (void*)&d_tris, &base, (void*)&n_nodes_dev};
            if ((r = launch(c, c->k->emit, n_blocks, DCSG_BLOCK, 0, s, args))) return r;
        }
        st.triangles[level] = n_new_tris;
        n_tris += n_new_tris;
        n_nodes = n_children;
        deferred = defer;
        cur ^= 1;
        for (int k = 0; k < 3; k++) hd[k] = 0.5f * hd[k];
    }
    unsigned long long* ls = (unsigned long long*)c->h_words;
    HIPCHK(c, hipMemcpyAsync(ls, d_lstats, DCSG_MAX_LEVELS * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int l = 0; l < DCSG_MAX_LEVELS; l++) {
        st.culled[l] = ls[4 * l];
        st.subdivided[l] = ls[4 * l + 1];
        st.leaves[l] = ls[4 * l + 2];
        st.triangles[l] = ls[4 * l + 3];
        st.nodes[l] = st.culled[l] + st.subdivided[l] + st.leaves[l];
    }
    c->n_tris = n_tris;
    if (n_triangles) *n_triangles = n_tris;
    if (stats) *stats = st;
    return 0;
}
}  // namespace

extern "C" {

// Mesh::getSurface, mesh.hpp:82-389, level-synchronous on the device.
int dcsg_mesh_lattice_device(dcsg_ctx* c, const float center[3], float diameter, const dcsg_export_config* cfg,
                             const float* d_sdf, const float* d_normals, int z_lo, int z_hi, int z_own_lo,
                             int z_own_hi, uint64_t* n_triangles, dcsg_mesh_stats* stats, void* stream_v)
{
    return mesh_impl(c, center, diameter, cfg, d_sdf, d_normals, z_lo, z_hi, z_own_lo, z_own_hi, n_triangles, stats, stream_v,
                     nullptr);
}

}  // extern "C"

namespace {
// The cube the cell lists cover: every object's bounding sphere (its brush lives in [-0.5, 0.5]^3 of the local frame,
// whose axes are the reciprocals of the stored vectors, scenecompiler.py:538-541), a twentieth more.  Vertices outside
// it are still projected correctly (the pseudo-cell of the whole program), only slowly.
DcsgCellGrid cell_grid_for(const Scene& sc, int g)
{
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool any = false;
    for (int o = 0; o < sc.n_objects; o++) {
        if (sc.shape[o] <= 1) continue;  // null / void brushes have no surface
        double r2 = 0;
        const float* ax[3] = {&sc.right[3 * o], &sc.up[3 * o], &sc.fwd[3 * o]};
        for (auto a : ax) {
            const double q = (double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2];
            r2 += q > 0 ? 0.25 / q : 1e12;  // half an axis: 0.5 / |reciprocal vector|
        }
        const float rad = (float)std::sqrt(r2);
        if (!(rad < 1e5f)) continue;
        for (int k = 0; k < 3; k++) {
            const float a = sc.pos[3 * o + k] - rad, b = sc.pos[3 * o + k] + rad;
            lo[k] = any ? std::min(lo[k], a) : a;
            hi[k] = any ? std::max(hi[k], b) : b;
        }
        any = true;
    }
    float side = 1.0f;
    for (int k = 0; k < 3; k++) side = std::max(side, hi[k] - lo[k]);
    side *= 1.05f;
    DcsgCellGrid G;
    G.g = g;
    G.cell = side / (float)g;
    G.inv_cell = (float)g / side;
    G.x0 = 0.5f * (lo[0] + hi[0]) - 0.5f * side;
    G.y0 = 0.5f * (lo[1] + hi[1]) - 0.5f * side;
    G.z0 = 0.5f * (lo[2] + hi[2]) - 0.5f * side;
    return G;
}

// Device words of the cell-list machinery (their own buffer: the entry counter lives as long as the scene's cells do,
// and the exports clear d_small): 0 entry counter, 1 vertex queue, 2 parked count, 3 cells built
unsigned* cells_word(dcsg_ctx* c, int k) { return (unsigned*)c->d_cell_words.p + k; }

// Grid and empty tables for the loaded scene (no cell built yet).
int ensure_cell_grid(dcsg_ctx* c, hipStream_t s)
{
    if (c->cells_valid) return 0;
    const char* eg = getenv("DCSG_CELLS");
    int g = eg ? atoi(eg) : 128;  // measured 32 .. 256 on the 256-brush scenes at 512^3 and 1024^3: 128 is the fastest, cold and warm
    g = (g < 4 ? 4 : (g > 256 ? 256 : g)) & ~3;  // a coarse cell is 4 x 4 x 4 fine ones
    c->cell_grid = cell_grid_for(c->scene, g);
    const uint64_t n_cells = (uint64_t)g * g * g, n_coarse = n_cells / 64, n_all = n_cells + 1 + n_coarse;
    int r;
    if ((r = ensure(c, c->d_cells, n_all * sizeof(uint32_t) * 2))) return r;
    if ((r = ensure(c, c->d_cell_want, n_all))) return r;
    if ((r = ensure(c, c->d_cell_words, 256))) return r;
    if ((r = ensure(c, c->d_cell_work, (n_all + 1) * sizeof(uint32_t)))) return r;
    // entries are handed out in chunks of 512 per wave of the build (kernels.hip DCSG_CELL_CHUNK): room for the lists and for
    // a part-used chunk per wave of a launch
    uint64_t capacity = std::max<uint64_t>(n_cells * 8, 4u << 20) + (uint64_t)c->n_cu * 12 * 4 * 512 + (uint64_t)c->scene.n_stack_prog;
    if (const char* e = getenv("DCSG_CELLS_CAPACITY")) capacity = (uint64_t)atoll(e);  // tests: start small, exercise the growth
    if ((r = ensure(c, c->d_cell_entries, capacity * sizeof(uint32_t)))) return r;
    HIPCHK(c, hipMemsetAsync(c->d_cells.p, 0xff, n_all * sizeof(uint32_t) * 2, s));  // DCSG_CELL_NOT_BUILT
    HIPCHK(c, hipMemsetAsync(c->d_cell_want.p, 0, n_cells, s));
    // wanted from the start: the pseudo-cell of the whole program and every coarse cell
    HIPCHK(c, hipMemsetAsync((char*)c->d_cell_want.p + n_cells, 1, 1 + n_coarse, s));
    HIPCHK(c, hipMemsetAsync(cells_word(c, 0), 0, 4 * sizeof(unsigned), s));
    c->cell_entries = 0;
    c->cells_built = 0;
    c->cells_built_ms = 0;
    c->cells_valid = true;
    return 0;
}

// Builds every wanted cell.  The entry buffer grows when the counter says it was too small (the cells that did not fit stay
// wanted and are built by the next launch).
int build_wanted_cells(dcsg_ctx* c, hipStream_t s)
{
    const uint64_t n_fine = (uint64_t)c->cell_grid.g * c->cell_grid.g * c->cell_grid.g, n_cells = n_fine + n_fine / 64;  // + the pseudo-cell below
    const double t0 = now_s();
    int r;
    for (int attempt = 0; attempt < 8; attempt++) {
        const uint64_t capacity = c->d_cell_entries.cap / sizeof(uint32_t);
        const void* blob = c->d_blob.p;
        int nv = c->scene.blob_vec4();
        void* cells = c->d_cells.p;
        void* entries = c->d_cell_entries.p;
        void* want = c->d_cell_want.p;
        unsigned* counter = cells_word(c, 0);
        unsigned cap = (unsigned)std::min<uint64_t>(capacity, 0xfffffff0ull);
        // the wanted cells as a list (they are clustered), then one wave per list entry
        void* work = c->d_cell_work.p;
        HIPCHK(c, hipMemsetAsync(work, 0, sizeof(unsigned), s));
        {
            unsigned np1 = (unsigned)(n_cells + 1);
            void* cargs[] = {&want, &np1, &work};
            if ((r = launch(c, c->kb.cells_collect, (unsigned)std::min<uint64_t>((n_cells + 256) / 256, (uint64_t)c->n_cu * 8), DCSG_BLOCK, 0, s, cargs)))
                return r;
        }
        void* args[] = {(void*)&blob, &nv, (void*)&c->cell_grid, &cells, &entries, (void*)&counter, &cap, &want, &work};
        const unsigned blocks = (unsigned)std::min<uint64_t>((n_cells + 4) / 4, (uint64_t)c->n_cu * 12);
        if ((r = launch(c, c->kb.cells_build, blocks, DCSG_BLOCK, c->scene.lds_bytes(), s, args))) return r;
        unsigned* words = (unsigned*)c->h_words;
        HIPCHK(c, hipMemcpyAsync(words, counter, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const unsigned used = words[0];
        c->cells_built = words[3];
        if (used <= cap) {
            c->cell_entries = used;
            c->cells_built_ms += (now_s() - t0) * 1e3;
            return 0;
        }
        // too small: the counter ran past the capacity.  Entries written so far stay where they are (cells point at them);
        // the counter goes back to the last entry that was written and the buffer grows, keeping its contents
        if (cap >= 0xfffffff0u) return fail(c, DCSG_ERR_INVALID, "cell lists do not fit 2^32 entries");
        const uint64_t want_cap = (uint64_t)used * 2 + 4096;
        // the cells that fitted were appended below `cap`; everything in [0, cap) is live or a hole
        if ((r = ensure(c, c->d_cell_entries, want_cap * sizeof(uint32_t), true, (size_t)cap * sizeof(uint32_t)))) return r;
        HIPCHK(c, hipMemcpyAsync(counter, &cap, sizeof(unsigned), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return fail(c, DCSG_ERR_HIP, "cell lists: the entry count kept growing");
}

// The grid and the cells every scene has from the start: the coarse cells and the pseudo-cell of the whole program
int ensure_coarse_cells(dcsg_ctx* c, hipStream_t s)
{
    const bool fresh = !c->cells_valid;
    int r = ensure_cell_grid(c, s);
    if (r) return r;
    return fresh ? build_wanted_cells(c, s) : 0;  // they are wanted from the start
}


// Projection through the cell lists: many-object scenes of built-in brushes (variant 4).  The cells the vertices start in
// (and their neighbours) are built first, together with whatever earlier projections of this scene asked for; a vertex that
// strays into a cell nobody built reads the coarse cell around it and leaves a request.  One launch, no read-back besides
// the build's entry counter.
int project_cells(dcsg_ctx* c, float* d_xyz, unsigned n, int steps, hipStream_t s)
{
    int r = ensure_cell_grid(c, s);
    if (r) return r;
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    void* want = c->d_cell_want.p;
    {
        void* args[] = {(void*)&d_xyz, &n, (void*)&c->cell_grid, &want};
        if ((r = launch(c, c->kb.cells_mark, sweep_grid(c, (n + DCSG_BLOCK - 1) / DCSG_BLOCK), DCSG_BLOCK, 0, s, args))) return r;
    }
    if ((r = build_wanted_cells(c, s))) return r;
    const int refill_env = getenv("DCSG_PROJECT_REFILL") ? atoi(getenv("DCSG_PROJECT_REFILL")) : 16;
    int refill_at = refill_env < 1 ? 1 : (refill_env > 64 ? 64 : refill_env);
    const int per_cu = getenv("DCSG_PROJECT_BLOCKS_PER_CU") ? atoi(getenv("DCSG_PROJECT_BLOCKS_PER_CU")) : 4;
    unsigned* queue = cells_word(c, 1);
    HIPCHK(c, hipMemsetAsync(queue, 0, sizeof(unsigned), s));
    const void* cells = c->d_cells.p;
    const void* entries = c->d_cell_entries.p;
    void* args[] = {(void*)&blob, &nv, (void*)&d_xyz, &n, &steps, (void*)&c->cell_grid, (void*)&cells, (void*)&entries, (void*)&queue,
                    &refill_at, &want};
    // persistent: as many blocks as stay resident (4 waves per SIMD), never more than there are 256-vertex groups
    unsigned blocks = (unsigned)c->n_cu * (unsigned)(per_cu < 1 ? 1 : per_cu);
    const unsigned groups = (n + DCSG_BLOCK - 1) / DCSG_BLOCK;
    if (blocks > groups) blocks = groups;
    // the kernel reads the scene's tables from LDS and nothing else of the scene's area (no interpreter stack, no list area)
    return launch(c, c->kb.project_cells, blocks, DCSG_BLOCK, (size_t)nv * 16, s, args);
}
}  // namespace

extern "C" {

// Diagnostic, no device needed: the cut planes dcsg_export_slab derives from a stage-A histogram (256 counts of near-surface
// probes per probe plane z = half_neg + cd (k - 128)) for a box of `diameter` centred at center_z: cuts[0 .. world] in
// lattice planes (cuts[r] .. cuts[r + 1] is rank r's slab).  Returns 0, or DCSG_ERR_INVALID when the equal slabs would be
// kept (sweep level coarser than the slabs, empty histogram: cuts then hold the equal slabs).
int dcsg_debug_slab_cuts(const uint32_t hist[256], float cd, float half_neg, float center_z, float diameter, int min_level, int grid_level,
                         int world, int* cuts)
{
    if (!hist || !cuts || world < 1 || grid_level < 1 || grid_level > 12 || ((1 << grid_level) % world) != 0) return DCSG_ERR_INVALID;
    dcsg_ctx tmp;
    memcpy(tmp.probe_z_hist, hist, sizeof(tmp.probe_z_hist));
    tmp.probe_cd = cd;
    tmp.probe_half_neg = half_neg;
    dcsg_export_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.min_level = min_level;
    cfg.grid_level = grid_level;
    const int res = 1 << grid_level;
    bool balanced = false;
    for (int r = 0; r < world; r++) {
        int lo = r * (res / world), hi = (r + 1) * (res / world);
        balanced = slab_cut_planes(&tmp, cfg, center_z, diameter, r, world, lo, hi);
        cuts[r] = lo;
        cuts[r + 1] = hi;
    }
    return balanced ? 0 : DCSG_ERR_INVALID;
}

int dcsg_debug_cells(const dcsg_ctx* c, int* grid, uint64_t* cells_built, uint64_t* entries, double* build_ms)
{
    if (!c) return DCSG_ERR_INVALID;
    if (grid) *grid = c->cells_valid ? c->cell_grid.g : 0;
    if (cells_built) *cells_built = c->cells_built;
    if (entries) *entries = c->cell_entries;
    if (build_ms) *build_ms = c->cells_built_ms;
    return 0;
}

int dcsg_project_vertices_device(dcsg_ctx* c, float* d_xyz, size_t n_vertices, int steps, void* stream)
{
    int r = need_scene(c);
    if (r) return r;
    if (steps <= 0 || n_vertices == 0) return 0;
    if (n_vertices > 0xffffff00ull) return fail(c, DCSG_ERR_INVALID, "too many vertices for one launch");
    if (c->variant == 4) return project_cells(c, d_xyz, (unsigned)n_vertices, steps, (hipStream_t)stream);
    const void* blob = c->d_blob.p;
    int nv = c->scene.blob_vec4();
    unsigned n = (unsigned)n_vertices;
    unsigned tiles = (n + DCSG_BLOCK - 1) / DCSG_BLOCK;
    void* args[] = {(void*)&blob, &nv, (void*)&d_xyz, &n, &steps};
    if (c->variant == 3 && c->k->project_points && !c->light_scene && !getenv("DCSG_PROJECT_VERTEX_PER_LANE")) {
        // per-scene code: eight vertices per wave, a lane per stencil point (32 vertices per block).  Kernel time of an
        // export's projection, vertex per lane -> this: Design2 512^3 2.33 -> 1.10 ms, 1024^3 4.40 -> 2.80; not for scenes of
        // a few standard brushes, whose evaluation is too small to be worth seven lanes (sphere 512^3 0.28 -> 0.46)
        tiles = (n + DCSG_BLOCK / 8 - 1) / (DCSG_BLOCK / 8);
        return launch(c, c->k->project_points, sweep_grid(c, tiles), DCSG_BLOCK, 0, (hipStream_t)stream, args);
    }
    return launch(c, c->k->project[c->variant], sweep_grid(c, tiles), DCSG_BLOCK, c->scene.lds_bytes(), (hipStream_t)stream, args);
}

}  // extern "C"

namespace {
// Vertex de-duplication up to the compact list (kernels.hip): owners of distinct positions in buffer order.  After it
// returns, compact[slot] holds the positions, slot_of / rep map every vertex to its slot, and the scanned per-block owner
// counts (block_sums[2*b]) give the number of owners before vertex 256 b.  probe_blocks / probe_out: scanned counts to
// fetch along with the total (chunk boundaries of the pipelined export).
struct Dedup {
    unsigned n = 0, n_blocks = 0, n_unique = 0;
};
int dedup_prepare(dcsg_ctx* c, hipStream_t s, Dedup& d, const unsigned* probe_blocks = nullptr, unsigned* probe_out = nullptr,
                  int n_probe = 0)
{
    const size_t n_verts = (size_t)c->n_tris * 3;
    int r;
    unsigned n = (unsigned)n_verts, n_blocks = (n + DCSG_BLOCK - 1) / DCSG_BLOCK;
    uint64_t table_size64 = 1;
    while (table_size64 < 2ull * n) table_size64 <<= 1;  // <= 2^31 for n <= 2^30
    unsigned table_size = (unsigned)table_size64;
    unsigned mask = table_size - 1;
    if ((r = ensure(c, c->d_dd_table, (size_t)table_size * 4))) return r;
    if ((r = ensure(c, c->d_dd_rep, (size_t)n * 4))) return r;
    if ((r = ensure(c, c->d_dd_slot, (size_t)n * 4))) return r;
    if ((r = ensure(c, c->d_block_sums, (size_t)n_blocks * 2 * sizeof(unsigned)))) return r;
    if ((r = ensure(c, c->d_small, 4096))) return r;
    HIPCHK(c, hipMemsetAsync(c->d_dd_table.p, 0xff, (size_t)table_size * 4, s));
    float* verts = (float*)c->d_tris.p;
    unsigned* table = (unsigned*)c->d_dd_table.p;
    unsigned* rep = (unsigned*)c->d_dd_rep.p;
    unsigned* slot = (unsigned*)c->d_dd_slot.p;
    unsigned* bs = (unsigned*)c->d_block_sums.p;
    unsigned* totals = (unsigned*)c->d_small.p;
    unsigned grid = sweep_grid(c, n_blocks);
    {
        void* args[] = {(void*)&verts, &n, (void*)&table, &mask, (void*)&rep};
        if ((r = launch(c, c->k->dd_insert, grid, DCSG_BLOCK, 0, s, args))) return r;
    }
    {
        void* args[] = {(void*)&table, &n, (void*)&rep};
        if ((r = launch(c, c->k->dd_resolve, grid, DCSG_BLOCK, 0, s, args))) return r;
    }
    {
        void* args[] = {(void*)&rep, &n, (void*)&bs};
        if ((r = launch(c, c->k->dd_count, n_blocks, DCSG_BLOCK, 0, s, args))) return r;
    }
    {
        const void* no_stats = nullptr;
        void* args[] = {(void*)&bs, &n_blocks, (void*)&totals, (void*)&no_stats, (void*)&no_stats};
        if ((r = launch(c, c->k->scan, 1, 1024, 0, s, args))) return r;
    }
    unsigned* tot = (unsigned*)c->h_words;  // pinned; the probes behind it (n_probe <= 8)
    HIPCHK(c, hipMemcpyAsync(tot, totals, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    for (int k = 0; k < n_probe; k++)
        HIPCHK(c, hipMemcpyAsync(tot + 2 + k, bs + 2 * (size_t)probe_blocks[k], sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int k = 0; k < n_probe; k++) probe_out[k] = tot[2 + k];
    unsigned n_unique = tot[0];
    if ((r = ensure(c, c->d_dd_compact, (size_t)(n_unique ? n_unique : 1) * 3 * sizeof(float)))) return r;
    float* compact = (float*)c->d_dd_compact.p;
    {
        void* args[] = {(void*)&verts, (void*)&rep, &n, (void*)&bs, (void*)&compact, (void*)&slot};
        if ((r = launch(c, c->k->dd_gather, n_blocks, DCSG_BLOCK, 0, s, args))) return r;
    }
    d.n = n;
    d.n_blocks = n_blocks;
    d.n_unique = n_unique;
    c->last_unique_vertices = n_unique;
    return 0;
}

// every copy in [first, first + count) takes the projected position of its representative
int dedup_scatter(dcsg_ctx* c, hipStream_t s, unsigned first, unsigned count)
{
    if (!count) return 0;
    float* verts = (float*)c->d_tris.p;
    unsigned* rep = (unsigned*)c->d_dd_rep.p;
    unsigned* slot = (unsigned*)c->d_dd_slot.p;
    float* compact = (float*)c->d_dd_compact.p;
    void* args[] = {(void*)&verts, (void*)&rep, (void*)&slot, (void*)&compact, &first, &count};
    return launch(c, c->k->dd_scatter, sweep_grid(c, (count + DCSG_BLOCK - 1) / DCSG_BLOCK), DCSG_BLOCK, 0, s, args);
}

bool dedup_applies(const dcsg_ctx* c, int steps)
{
    const size_t n_verts = (size_t)c->n_tris * 3;
    const char* env = getenv("DCSG_DEDUP");
    // the hash table has 2^k >= 2 n slots of 32-bit vertex ids: beyond 2^30 vertices the plain projection runs
    return !(steps <= 0 || n_verts < 4096 || n_verts > (1ull << 30) || (env && atoi(env) == 0));
}
}  // namespace

extern "C" {

// performGradientDescent on the context's triangle buffer.  Copies of one position move identically, so one
// representative per distinct position is projected (kernels.hip, "Vertex de-duplication") and copied back.
int dcsg_project_triangles(dcsg_ctx* c, int steps, void* stream)
{
    if (!c) return DCSG_ERR_INVALID;
    if (!dedup_applies(c, steps)) return dcsg_project_vertices_device(c, (float*)c->d_tris.p, (size_t)c->n_tris * 3, steps, stream);
    int r = need_scene(c);
    if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    Dedup d;
    if ((r = dedup_prepare(c, s, d))) return r;
    if ((r = dcsg_project_vertices_device(c, (float*)c->d_dd_compact.p, d.n_unique, steps, stream))) return r;
    return dedup_scatter(c, s, 0, d.n);
}

// Opt-in: T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529; kernels.hip, oracle_retopologize) on
// the context's triangle buffer: mesh vertices that lie on a triangle's edges become vertices of its polygon.
int dcsg_retopologize(dcsg_ctx* c, const float center[3], float diameter, int min_level, int grid_level, void* stream)
{
    int r = need_scene(c);
    if (r) return r;
    if (!center || grid_level < 1 || grid_level > 12 || min_level < 0 || min_level > grid_level || grid_level - min_level > 5)
        return fail(c, DCSG_ERR_INVALID, "retopologize needs 0 <= gridLevel - minimumOctreeLevel <= 5 and gridLevel <= 12");
    if (c->n_tris == 0) return 0;
    if (c->n_tris * 3 > (1ull << 30)) return fail(c, DCSG_ERR_INVALID, "too many vertices for the cell set");
    if ((r = need_rest(c))) return r;
    hipStream_t s = (hipStream_t)stream;
    DcsgRetopo R;
    for (int k = 0; k < 3; k++) {
        R.c[k] = center[k];
        R.hd[k] = diameter / 2.0f;  // DesignCSG.cpp:718
    }
    R.grid_points = (float)((1 << grid_level) + 1);
    R.points_along_edge = 1 << (grid_level - min_level);
    {   // joinDistance, mesh.hpp:445: double products narrowed to float arguments of min3, then 0.25 * in double
        float m = (float)(2.0 * R.hd[0] / R.grid_points);
        for (int k = 1; k < 3; k++) {
            float v = (float)(2.0 * R.hd[k] / R.grid_points);
            if (v < m) m = v;
        }
        volatile float jd = (float)(0.25 * m);
        R.join_distance2 = jd * jd;
    }
    const unsigned n_tris = (unsigned)c->n_tris, n_verts = n_tris * 3u, n_blocks = (n_tris + DCSG_BLOCK - 1) / DCSG_BLOCK;
    uint64_t cap = 64;
    while (cap < 2ull * n_verts) cap <<= 1;
    unsigned mask = (unsigned)(cap - 1);
    if ((r = ensure(c, c->d_dd_table, (size_t)cap * 12))) return r;  // keys, then occupants; shares the de-duplication table
    if ((r = ensure(c, c->d_rt_keep, (size_t)n_tris * 3 * 4))) return r;
    if ((r = ensure(c, c->d_block_sums, (size_t)n_blocks * 2 * sizeof(unsigned)))) return r;
    if ((r = ensure(c, c->d_small, 4096))) return r;
    unsigned long long* keys = (unsigned long long*)c->d_dd_table.p;
    unsigned* rep = (unsigned*)((char*)c->d_dd_table.p + (size_t)cap * 8);
    HIPCHK(c, hipMemsetAsync(keys, 0, (size_t)cap * 8, s));
    HIPCHK(c, hipMemsetAsync(rep, 0xff, (size_t)cap * 4, s));
    const float* tris = (const float*)c->d_tris.p;
    unsigned* keep = (unsigned*)c->d_rt_keep.p;
    unsigned* bs = (unsigned*)c->d_block_sums.p;
    unsigned* totals = (unsigned*)c->d_small.p;
    {
        void* args[] = {(void*)&tris, (void*)&n_verts, &R, (void*)&keys, (void*)&rep, &mask};
        if ((r = launch(c, c->k->rt_mark, sweep_grid(c, (n_verts + DCSG_BLOCK - 1) / DCSG_BLOCK), DCSG_BLOCK, 0, s, args))) return r;
    }
    {
        void* args[] = {(void*)&tris, (void*)&n_tris, &R, (void*)&keys, (void*)&rep, &mask, (void*)&keep, (void*)&bs};
        if ((r = launch(c, c->k->rt_count, n_blocks, DCSG_BLOCK, 0, s, args))) return r;
    }
    {
        const void* no_stats = nullptr;
        void* args[] = {(void*)&bs, (void*)&n_blocks, (void*)&totals, (void*)&no_stats, (void*)&no_stats};
        if ((r = launch(c, c->k->scan, 1, 1024, 0, s, args))) return r;
    }
    unsigned tot[2];
    HIPCHK(c, hipMemcpyAsync(tot, totals, sizeof(tot), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const unsigned n_out = tot[0];
    if ((r = ensure(c, c->d_rt_out, (size_t)(n_out ? n_out : 1) * 9 * sizeof(float)))) return r;
    float* out = (float*)c->d_rt_out.p;
    {
        void* args[] = {(void*)&tris, (void*)&n_tris, &R, (void*)&keys, (void*)&rep, &mask, (void*)&keep, (void*)&bs, (void*)&out};
        if ((r = launch(c, c->k->rt_emit, n_blocks, DCSG_BLOCK, 0, s, args))) return r;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    std::swap(c->d_tris, c->d_rt_out);
    c->n_tris = n_out;
    return 0;
}

int dcsg_triangles_device(dcsg_ctx* c, const float** d_triangles, uint64_t* n)
{
    if (!c) return DCSG_ERR_INVALID;
    if (d_triangles) *d_triangles = (const float*)c->d_tris.p;
    if (n) *n = c->n_tris;
    return 0;
}

// Replaces the context's triangle buffer with caller-provided triangles (9 floats each, host memory): lets the
// post-meshing stages (retopologize, projection, writers) run on any mesh.
int dcsg_upload_triangles(dcsg_ctx* c, const float* tris, uint64_t n)
{
    if (!c || (n && !tris)) return DCSG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int r = ensure(c, c->d_tris, (size_t)(n ? n : 1) * 9 * sizeof(float));
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCHK(c, hipMemcpy(c->d_tris.p, tris, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    c->n_tris = n;
    return 0;
}

int dcsg_download_triangles(dcsg_ctx* c, float* out, uint64_t max_triangles)
{
    if (!c) return DCSG_ERR_INVALID;
    if (max_triangles < c->n_tris) return fail(c, DCSG_ERR_INVALID, "output buffer too small");
    if (!c->n_tris) return 0;
    if (!out) return fail(c, DCSG_ERR_INVALID, "NULL output buffer");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_tris.p, (size_t)c->n_tris * 9 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int dcsg_copy_triangles_device(dcsg_ctx* c, float* d_out, uint64_t max_triangles, void* stream)
{
    if (!c) return DCSG_ERR_INVALID;
    if (max_triangles < c->n_tris) return fail(c, DCSG_ERR_INVALID, "output buffer too small");
    if (!c->n_tris) return 0;
    if (!d_out) return fail(c, DCSG_ERR_INVALID, "NULL output buffer");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_out, c->d_tris.p, (size_t)c->n_tris * 9 * sizeof(float), hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    return 0;
}

// File images are produced by a few host threads, each formatting its range of triangles into a small buffer
// and pwrite()-ing it at its final offset: formatting and the copies into the page cache run in parallel and no
// file-sized intermediate buffer exists.  dcsg_export additionally writes the two files concurrently.
}  // extern "C"

namespace {

// Runs f(t0, t1) on `nt` threads over [0, n) in equal ranges; returns false if any call returned false.
template <class F>
bool parallel_ranges(uint64_t n, unsigned nt, F f)
{
    if (n < (1u << 15)) nt = 1;
    std::vector<std::thread> th;
    std::vector<char> ok(nt, 1);
    for (unsigned t = 1; t < nt; t++) th.emplace_back([&, t] { ok[t] = f(n * t / nt, n * (t + 1) / nt) ? 1 : 0; });
    ok[0] = f(0, n / nt) ? 1 : 0;
    for (auto& x : th) x.join();
    for (char o : ok)
        if (!o) return false;
    return true;
}

bool pwrite_all(int fd, const unsigned char* buf, size_t bytes, off_t off)
{
    while (bytes) {
        ssize_t w = pwrite(fd, buf, bytes, off);
        if (w <= 0) return false;
        buf += w;
        off += w;
        bytes -= (size_t)w;
    }
    return true;
}

// An existing file is overwritten in place and cut to its new length at the end: truncating first makes the kernel
// free and re-allocate every page-cache page of the old file, which costs more than writing the bytes (C4: 36 ms
// against 12 ms for the 136 MB .ply).
int open_for_overwrite(const char* path) { return open(path, O_WRONLY | O_CREAT, 0644); }
// Reserve the final size before the parallel pwrites (a new file: block allocation once instead of per extension,
// C4 .ply 13.0 -> 10.5 ms on overlayfs).  fallocate(2) itself, not posix_fallocate: a file system without support
// answers EOPNOTSUPP and nothing is lost, where glibc's emulation would write every block.
bool reserve_file(int fd, off_t size)
{
    if (size <= 0 || getenv("DCSG_NO_FALLOCATE")) return false;
    return fallocate(fd, 0, 0, size) == 0;  // also extends the file to `size`
}
bool finish_file(int fd, off_t size, bool ok)
{
    if (ftruncate(fd, ok ? size : 0) != 0) ok = false;  // a failed write leaves an empty file, not a plausible one
    return close(fd) == 0 && ok;
}

// what a formatting thread formats before it writes it: small enough to still be in its cache when the kernel copies it
size_t writer_buffer_bytes()
{
    const char* e = getenv("DCSG_WRITER_BUFFER_KB");
    long kb = e ? atol(e) : 1280;
    return (size_t)(kb < 16 ? 16 : (kb > 65536 ? 65536 : kb)) << 10;
}

unsigned writer_threads()
{
    const char* e = getenv("DCSG_WRITER_THREADS");
    int n = e ? atoi(e) : 4;  // per file; the two files are written concurrently
    return n < 1 ? 1u : (n > 64 ? 64u : (unsigned)n);
}

// Crash safety of the in-place writers: a file is valid only between END and the next BEGIN.  BEGIN overwrites the
// first bytes with a marker no STL / PLY reader accepts as the start of a complete file, the payload follows, and only
// when every byte is in place is the file cut to its final length and given its real header (END).  A failed export
// leaves an empty file; an interrupted one a file that starts with the marker -- never a valid header over a stale or
// partial body.
static const char kIncomplete[] = "dcsg: incomplete export -- this file was being written when the exporter stopped\n";

int ply_header(char* header, size_t cap, uint64_t n)
{
    return snprintf(header, cap,
                    "ply\nformat binary_little_endian 1.0\n"
                    "comment Written with hapPLY (https://github.com/nmwsharp/happly)\n"
                    "element vertex %llu\nproperty double x\nproperty double y\nproperty double z\n"
                    "element face %llu\nproperty list uchar uint vertex_indices\nend_header\n",
                    (unsigned long long)(n * 3), (unsigned long long)n);
}

inline void stl_record(unsigned char* o, const float* v)
{
    float rec[12] = {0.0f, 0.0f, 0.0f, v[0], v[2], v[1], v[3], v[5], v[4], v[6], v[8], v[7]};
    memcpy(o, rec, 48);
    o[48] = o[49] = 0;
}

// (Rounds 2-4 carried a second path here that formatted straight into a shared mapping of the file.  It never ran: the
// files are opened write-only and mmap refuses a shared writable mapping of such a descriptor -- every number this repo
// ever reported came from the pwrite path below.  Measured for real this round (tools/fmt_probe.cpp, GPU box): first-touch
// faults of a shared file mapping cost 16-25 ms for the 136 MB .ply whatever the thread count, against 7-8 ms for these
// buffered pwrites, so the mapping is gone rather than repaired.)

// STL payload of triangles [first, first + n) of a file of `total` triangles: records at 84 + 50 t
bool stl_payload(int fd, const float* tris, uint64_t n, uint64_t first)
{
    return parallel_ranges(n, writer_threads(), [=](uint64_t t0, uint64_t t1) {
        const uint64_t step = writer_buffer_bytes() / 50;  // triangles per buffer
        std::vector<unsigned char> buf((size_t)step * 50);
        for (uint64_t a = t0; a < t1; a += step) {
            uint64_t b = a + step < t1 ? a + step : t1;
            unsigned char* o = buf.data();
            for (uint64_t t = a; t < b; t++, o += 50) stl_record(o, tris + t * 9);
            if (!pwrite_all(fd, buf.data(), (size_t)(b - a) * 50, (off_t)(84 + (first + a) * 50))) return false;
        }
        return true;
    });
}

// PLY face records of triangles [first, first + n) of a file of `total`: at hl + 72 total + 13 t.  They depend on nothing
// but t.
bool ply_faces(int fd, uint64_t n, uint64_t first, uint64_t total)
{
    char header[512];
    const off_t faces_at = ply_header(header, sizeof(header), total) + (off_t)(total * 72);
    return parallel_ranges(n, writer_threads(), [=](uint64_t t0, uint64_t t1) {
        const uint64_t step = writer_buffer_bytes() / 13;  // triangles per buffer
        std::vector<unsigned char> fbuf((size_t)step * 13);
        for (uint64_t a = t0; a < t1; a += step) {
            uint64_t b = a + step < t1 ? a + step : t1;
            unsigned char* q = fbuf.data();
            for (uint64_t t = first + a; t < first + b; t++, q += 13) {
                q[0] = 3;
                uint32_t idx[3] = {(uint32_t)(3 * t), (uint32_t)(3 * t + 1), (uint32_t)(3 * t + 2)};
                memcpy(q + 1, idx, 12);
            }
            if (!pwrite_all(fd, fbuf.data(), (size_t)(b - a) * 13, faces_at + (off_t)((first + a) * 13))) return false;
        }
        return true;
    });
}

// PLY vertices of triangles [first, first + n) of a file of `total`: at hl + 72 t
bool ply_vertices(int fd, const float* tris, uint64_t n, uint64_t first, uint64_t total)
{
    char header[512];
    const off_t verts_at = ply_header(header, sizeof(header), total);
    return parallel_ranges(n, writer_threads(), [=](uint64_t t0, uint64_t t1) {
        const uint64_t step = writer_buffer_bytes() / 72;  // triangles per buffer
        std::vector<unsigned char> vbuf((size_t)step * 72);
        for (uint64_t a = t0; a < t1; a += step) {
            uint64_t b = a + step < t1 ? a + step : t1;
            unsigned char* o = vbuf.data();
            for (uint64_t i = a * 9; i < b * 9; i++, o += 8) {
                double v = (double)tris[i];
                memcpy(o, &v, 8);
            }
            if (!pwrite_all(fd, vbuf.data(), (size_t)(b - a) * 72, verts_at + (off_t)((first + a) * 72))) return false;
        }
        return true;
    });
}

// One file format's BEGIN / payload / END (see dcsg_write_stl_range in dcsg.h)
int write_range(const char* path, const float* tris, uint64_t n, uint64_t first, uint64_t total, unsigned flags, bool ply)
{
    if (!path || (n && !tris) || first + n > total) return DCSG_ERR_INVALID;
    if (!ply && total > 0xffffffffull) return DCSG_ERR_INVALID;      // the STL header holds a 32-bit triangle count
    if (ply && total * 3 > 0xffffffffull) return DCSG_ERR_INVALID;   // face lists index vertices with 32-bit uints
    char header[512];
    int hl = 84;
    if (ply) {
        hl = ply_header(header, sizeof(header), total);
    } else {
        memset(header, 0, 84);
        uint32_t n32 = (uint32_t)total;
        memcpy(header + 80, &n32, 4);
    }
    const off_t size = (off_t)hl + (off_t)(total * (ply ? 85 : 50));
    const bool begin = (flags & DCSG_WRITE_BEGIN) != 0, end = (flags & DCSG_WRITE_END) != 0;
    int fd = begin ? open_for_overwrite(path) : open(path, O_WRONLY);
    if (fd < 0) return DCSG_ERR_IO;
    bool ok = true;
    if (begin) {
        ok = pwrite_all(fd, (const unsigned char*)kIncomplete, sizeof(kIncomplete) - 1, 0);
        reserve_file(fd, size);
    }
    if (ok && n) ok = ply ? (ply_vertices(fd, tris, n, first, total) && ply_faces(fd, n, first, total)) : stl_payload(fd, tris, n, first);
    if (ok && end) ok = ftruncate(fd, size) == 0 && pwrite_all(fd, (const unsigned char*)header, (size_t)hl, 0);
    if (!ok && ftruncate(fd, 0) != 0) ok = false;  // nothing that could pass for a finished file stays behind
    return close(fd) == 0 && ok ? 0 : DCSG_ERR_IO;
}

// One output file of a streamed export (stream_triangles_to_files), held open over all its pieces: BEGIN and the
// reservation happen once, each piece is formatted into small buffers and pwritten at its final offsets, END comes with
// close().  Calling write_range per piece instead re-opened and re-reserved the file every time, which on a 235 MB .ply
// cost more than a millisecond per piece (flat256 1024^3: .ply 12.8 -> 20.1 ms in six pieces).  Same bytes, same
// crash-safety protocol as write_range.
class FileSink {
public:
    // triangles [first, first + n) of a file of `total` will be written through this sink
    int open(const char* path, bool ply, uint64_t first, uint64_t n, uint64_t total, bool begin)
    {
        if (!path || first + n > total) return DCSG_ERR_INVALID;
        if (!ply && total > 0xffffffffull) return DCSG_ERR_INVALID;
        if (ply && total * 3 > 0xffffffffull) return DCSG_ERR_INVALID;
        ply_ = ply, first_ = first, total_ = total;
        if (ply) {
            hl_ = ply_header(header_, sizeof(header_), total);
        } else {
            memset(header_, 0, 84);
            uint32_t n32 = (uint32_t)total;
            memcpy(header_ + 80, &n32, 4);
        }
        size_ = (off_t)hl_ + (off_t)(total * (ply ? 85 : 50));
        fd_ = begin ? open_for_overwrite(path) : ::open(path, O_WRONLY);
        if (fd_ < 0) return DCSG_ERR_IO;
        if (begin) {
            ok_ = pwrite_all(fd_, (const unsigned char*)kIncomplete, sizeof(kIncomplete) - 1, 0);
            reserve_file(fd_, size_);
        }
        return 0;
    }
    // PLY: the face records of all n triangles of this sink -- (3; 3t, 3t+1, 3t+2) depends on nothing but t, so they are
    // written while the first piece of vertices is still being projected / downloaded (15 % of the file's bytes)
    void write_faces_ahead(uint64_t n)
    {
        if (!ok_ || !ply_ || !n || faces_done_) return;
        ok_ = ply_faces(fd_, n, first_, total_);
        faces_done_ = true;
    }
    // the piece `tris` = triangles [first + at, first + at + cnt)
    void write(const float* tris, uint64_t at, uint64_t cnt)
    {
        if (!ok_ || !cnt) return;
        if (ply_) ok_ = ply_vertices(fd_, tris, cnt, first_ + at, total_) && (faces_done_ || ply_faces(fd_, cnt, first_ + at, total_));
        else ok_ = stl_payload(fd_, tris, cnt, first_ + at);
    }
    int close(bool end)
    {
        if (fd_ < 0) return DCSG_ERR_IO;
        if (ok_ && end) ok_ = ftruncate(fd_, size_) == 0 && pwrite_all(fd_, (const unsigned char*)header_, (size_t)hl_, 0);
        if (!ok_ && ftruncate(fd_, 0) != 0) ok_ = false;
        const bool closed = ::close(fd_) == 0;
        fd_ = -1;
        return closed && ok_ ? 0 : DCSG_ERR_IO;
    }
    ~FileSink()
    {
        if (fd_ >= 0) {
            ok_ = false;
            close(false);
        }
    }

private:
    int fd_ = -1, hl_ = 84;
    bool ply_ = false, ok_ = true, faces_done_ = false;
    uint64_t first_ = 0, total_ = 0;
    off_t size_ = 0;
    char header_[512];
};

}  // namespace

extern "C" {

// writeTrianglesToSTL, utils.hpp:41-103: 80 zero bytes, u32 count, per triangle a zero normal, then
// A, B, C each written as (x, z, y), then u16 0.  Little-endian host assumed (x86-64).
int dcsg_write_stl(const char* path, const float* tris, uint64_t n)
{
    return write_range(path, tris, n, 0, n, DCSG_WRITE_BEGIN | DCSG_WRITE_END, false);
}

// writeTrianglesToPLY, utils.hpp:106-154 via happly.h:1998-2039 (header), 1538-1570 (double x,y,z),
// 1640-1670 (uchar count + uint indices): un-indexed, vertices as doubles, faces (3i,3i+1,3i+2).
int dcsg_write_ply(const char* path, const float* tris, uint64_t n)
{
    return write_range(path, tris, n, 0, n, DCSG_WRITE_BEGIN | DCSG_WRITE_END, true);
}

// Byte-range forms for sharded exports ("each rank writes its byte range of the STL/PLY"): both
// formats are fixed-size records once the total count is known.
int dcsg_write_stl_range(const char* path, const float* tris, uint64_t n, uint64_t first, uint64_t total, unsigned flags)
{
    return write_range(path, tris, n, first, total, flags, false);
}
int dcsg_write_ply_range(const char* path, const float* tris, uint64_t n, uint64_t first, uint64_t total, unsigned flags)
{
    return write_range(path, tris, n, first, total, flags, true);
}

// ---- opt-in writers beyond the reference's -----------------------------------------------
// STL with real facet normals: same records as dcsg_write_stl, the normal = normalize(cross(B-A, C-A)) of the
// vertices AS WRITTEN (x, z, y), in float, left to right; degenerate facets keep the zero normal.
int dcsg_write_stl_normals(const char* path, const float* tris, uint64_t n)
{
    if (!path || (n && !tris)) return DCSG_ERR_INVALID;
    if (n > 0xffffffffull) return DCSG_ERR_INVALID;
    int fd = open_for_overwrite(path);
    if (fd < 0) return DCSG_ERR_IO;
    reserve_file(fd, (off_t)(84 + n * 50));
    unsigned char head[84];
    memset(head, 0, sizeof(head));
    uint32_t n32 = (uint32_t)n;
    memcpy(head + 80, &n32, 4);
    bool ok = pwrite_all(fd, head, sizeof(head), 0);
    ok = ok && parallel_ranges(n, writer_threads(), [=](uint64_t t0, uint64_t t1) {
        const uint64_t step = 1u << 15;
        std::vector<unsigned char> buf((size_t)step * 50);
        for (uint64_t a = t0; a < t1; a += step) {
            uint64_t b = a + step < t1 ? a + step : t1;
            unsigned char* o = buf.data();
            for (uint64_t t = a; t < b; t++, o += 50) {
                const float* v = tris + t * 9;
                float rec[12] = {0.0f, 0.0f, 0.0f, v[0], v[2], v[1], v[3], v[5], v[4], v[6], v[8], v[7]};
                float ux = rec[6] - rec[3], uy = rec[7] - rec[4], uz = rec[8] - rec[5];
                float wx = rec[9] - rec[3], wy = rec[10] - rec[4], wz = rec[11] - rec[5];
                float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
                float len = sqrtf(nx * nx + ny * ny + nz * nz);
                if (len > 0.0f && len - len == 0.0f) {
                    rec[0] = nx / len;
                    rec[1] = ny / len;
                    rec[2] = nz / len;
                }
                memcpy(o, rec, 48);
                o[48] = o[49] = 0;
            }
            if (!pwrite_all(fd, buf.data(), (size_t)(b - a) * 50, (off_t)(84 + a * 50))) return false;
        }
        return true;
    });
    return finish_file(fd, (off_t)(84 + n * 50), ok) ? 0 : DCSG_ERR_IO;
}

// Indexed PLY: every distinct vertex position (bit pattern) once, in order of first appearance, as float
// x y z; faces index them.  Same axis order as dcsg_write_ply (x, y, z as stored, no swap).
int dcsg_write_ply_indexed(const char* path, const float* tris, uint64_t n, uint64_t* n_unique_out)
{
    if (!path || (n && !tris)) return DCSG_ERR_INVALID;
    if (n * 3 >= 0xffffffffull) return DCSG_ERR_INVALID;
    // open-addressing table over the 96-bit position; value = vertex id
    uint64_t cap = 16;
    while (cap < n * 3 * 2) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0xffffffffu);
    std::vector<uint32_t> index(n * 3);
    std::vector<float> verts;
    verts.reserve((size_t)(n * 3 / 4 + 16) * 3);
    for (uint64_t i = 0; i < n * 3; i++) {
        uint32_t w[3];
        memcpy(w, tris + i * 3, 12);
        uint64_t h = fnv1a64(w, 12, 0x9e3779b97f4a7c15ull) & (cap - 1);
        for (;;) {
            uint32_t id = slot[h];
            if (id == 0xffffffffu) {
                id = (uint32_t)(verts.size() / 3);
                slot[h] = id;
                verts.insert(verts.end(), tris + i * 3, tris + i * 3 + 3);
                index[i] = id;
                break;
            }
            if (memcmp(&verts[(size_t)id * 3], w, 12) == 0) {
                index[i] = id;
                break;
            }
            h = (h + 1) & (cap - 1);
        }
    }
    const uint64_t nv = verts.size() / 3;
    if (n_unique_out) *n_unique_out = nv;
    char header[512];
    int hl = snprintf(header, sizeof(header),
                      "ply\nformat binary_little_endian 1.0\ncomment designcsg-amd indexed mesh\n"
                      "element vertex %llu\nproperty float x\nproperty float y\nproperty float z\n"
                      "element face %llu\nproperty list uchar uint vertex_indices\nend_header\n",
                      (unsigned long long)nv, (unsigned long long)n);
    int fd = open_for_overwrite(path);
    if (fd < 0) return DCSG_ERR_IO;
    reserve_file(fd, (off_t)hl + (off_t)(nv * 12 + n * 13));
    bool ok = pwrite_all(fd, (const unsigned char*)header, (size_t)hl, 0);
    ok = ok && pwrite_all(fd, (const unsigned char*)verts.data(), (size_t)nv * 12, hl);
    const off_t faces_at = hl + (off_t)(nv * 12);
    const uint32_t* idx = index.data();
    ok = ok && parallel_ranges(n, writer_threads(), [=](uint64_t t0, uint64_t t1) {
        const uint64_t step = 1u << 16;
        std::vector<unsigned char> fbuf((size_t)step * 13);
        for (uint64_t a = t0; a < t1; a += step) {
            uint64_t b = a + step < t1 ? a + step : t1;
            unsigned char* q = fbuf.data();
            for (uint64_t t = a; t < b; t++, q += 13) {
                q[0] = 3;
                memcpy(q + 1, idx + 3 * t, 12);
            }
            if (!pwrite_all(fd, fbuf.data(), (size_t)(b - a) * 13, faces_at + (off_t)(a * 13))) return false;
        }
        return true;
    });
    return finish_file(fd, (off_t)hl + (off_t)(nv * 12 + n * 13), ok) ? 0 : DCSG_ERR_IO;
}

int dcsg_debug_sqrt_device(dcsg_ctx* c, const float* d_in, size_t n, float* d_out, void* stream)
{
    if (!c || n > 0xffffff00ull) return DCSG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned cnt = (unsigned)n;
    void* args[] = {(void*)&d_in, (void*)&d_out, &cnt};
    if (c->scene_loaded)
        if (int rr = need_rest(c)) return rr;
    hipFunction_t f = (c->scene_loaded && c->jit) ? c->ks.test_sqrt : c->kb.test_sqrt;
    return launch(c, f, (unsigned)c->n_cu * 16u, DCSG_BLOCK, 0, (hipStream_t)stream, args);
}

int dcsg_debug_normal_device(dcsg_ctx* c, const float* d_in, size_t n, float dy, float dz, int mixed, float* d_out, void* stream)
{
    if (!c || n > 0x55555500ull) return DCSG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned cnt = (unsigned)n;
    void* args[] = {(void*)&d_in, (void*)&d_out, &cnt, &dy, &dz, &mixed};
    if (c->scene_loaded)
        if (int rr = need_rest(c)) return rr;
    hipFunction_t f = (c->scene_loaded && c->jit) ? c->ks.test_normal : c->kb.test_normal;
    return launch(c, f, (unsigned)c->n_cu * 16u, DCSG_BLOCK, 0, (hipStream_t)stream, args);
}

// Diagnostic, no device needed (thread-sanitizer harness, tests/test_host_threads.py): the host-side sequence of a cold
// dcsg_load_scene -- export part compiled on this thread, rest part handed to the background job -- on a context that never
// touches HIP.  *slot holds that context between calls (NULL: a new one).  mode 0: wait for the background build and report
// it; mode 1: return at once with the build still running, as an early error return of dcsg_load_scene would (the next
// call on the slot, or mode 2, has to cope); mode 2: destroy the slot's context (joins).  Returns 0, or DCSG_ERR_BUILD with
// the compiler's log in log_out.
int dcsg_debug_background_build(void** slot, const char* scene_txt, const char* bp_txt, const char* scene_cl, int mode, char* log_out,
                                size_t log_cap)
{
    if (!slot) return DCSG_ERR_INVALID;
    dcsg_ctx* c = (dcsg_ctx*)*slot;
    auto say = [&](const std::string& m) {
        if (log_out && log_cap) {
            strncpy(log_out, m.c_str(), log_cap - 1);
            log_out[log_cap - 1] = '\0';
        }
    };
    if (mode == 2) {
        delete c;  // ~RestJob joins
        *slot = nullptr;
        return 0;
    }
    if (!c) *slot = c = new dcsg_ctx();
    Scene s;
    std::string err, hsaco, log;
    ModulePlan plan;
    if (!parse_scene(scene_txt, bp_txt, s, err) || !plan_scene_module(s, scene_cl, plan, err)) {
        say(err);
        return DCSG_ERR_INVALID;
    }
    if (!plan.needs_jit) return 0;
    join_rest_thread(c);  // what every path of dcsg_load_scene does before it touches the job
    bool cached = false;
    if (!jit_scene_module(plan.cl, plan.extra, hsaco, log, &cached, plan.relaxed, plan.specialise, JIT_PART_EXPORT)) {
        say(log);
        return DCSG_ERR_BUILD;
    }
    c->rest_job.reset();
    start_rest_job(c, plan);
    if (mode == 1) return 0;
    join_rest_thread(c);
    if (!c->rest_job->ok) {
        say(c->rest_job->log);
        return DCSG_ERR_BUILD;
    }
    return 0;
}

int dcsg_jit_prebuild_scene(const char* scene_txt, const char* bp_txt, const char* scene_cl, char* log_out, size_t log_cap,
                             int* from_cache)
{
    return dcsg_jit_prebuild_scene_numerics(scene_txt, bp_txt, scene_cl, DCSG_NUMERICS_EXACT, log_out, log_cap, from_cache);
}

int dcsg_jit_prebuild_scene_numerics(const char* scene_txt, const char* bp_txt, const char* scene_cl, int numerics,
                                     char* log_out, size_t log_cap, int* from_cache)
{
    Scene s;
    std::string err, hsaco, log;
    ModulePlan plan;
    bool cached = false;
    int rc = 0;
    if (!parse_scene(scene_txt, bp_txt, s, err) || !plan_scene_module(s, scene_cl, plan, err, numerics == DCSG_NUMERICS_RELAXED)) {
        log = err;
        rc = DCSG_ERR_INVALID;
    } else if (plan.needs_jit) {
        cached = true;
        for (int part = 0; part < 2 && rc == 0; part++) {
            bool c1 = false;
            if (!jit_scene_module(plan.cl, plan.extra, hsaco, log, &c1, plan.relaxed, plan.specialise, part)) rc = DCSG_ERR_BUILD;
            cached = cached && c1;
        }
    }
    if (from_cache) *from_cache = cached ? 1 : 0;
    if (log_out && log_cap) {
        strncpy(log_out, log.c_str(), log_cap - 1);
        log_out[log_cap - 1] = '\0';
    }
    return rc;
}

float dcsg_angle_q_star(float complex_surface_threshold) { return angle_threshold_to_q_star(complex_surface_threshold); }

int dcsg_compile_program(const char* scene_txt, const char* bp_txt, int32_t* reg_prog, int* n_reg_steps, int* n_regs,
                         int32_t* stack_prog, int* n_stack_steps, int* stack_depth, int cap)
{
    Scene s;
    std::string err;
    if (!parse_scene(scene_txt, bp_txt, s, err)) return fail(nullptr, DCSG_ERR_INVALID, err);
    if (s.n_prog > cap || s.n_stack_prog > cap) return fail(nullptr, DCSG_ERR_INVALID, "output arrays too small");
    if (reg_prog && !s.prog.empty()) memcpy(reg_prog, s.prog.data(), s.prog.size() * sizeof(int32_t));
    if (stack_prog && !s.stack_prog.empty()) memcpy(stack_prog, s.stack_prog.data(), s.stack_prog.size() * sizeof(int32_t));
    if (n_reg_steps) *n_reg_steps = s.n_prog;
    if (n_regs) *n_regs = s.n_regs;
    if (n_stack_steps) *n_stack_steps = s.n_stack_prog;
    if (stack_depth) *stack_depth = s.stack_depth;
    return 0;
}

int dcsg_expand_lookup_table(const char* txt, uint8_t table[4096], int* total)
{
    std::vector<uint8_t> t;
    int n = 0;
    std::string err;
    if (!txt || !table) return DCSG_ERR_INVALID;
    if (!parse_lookup_table(txt, t, n, err)) return fail(nullptr, DCSG_ERR_INVALID, err);
    memcpy(table, t.data(), 4096);
    if (total) *total = n;
    return 0;
}

}  // extern "C"

namespace {

// Work-balanced Z slabs.  Where the work of an export is follows the surface: triangles, alive octree nodes, vertices to
// project.  Stage A has just counted, per probe plane in z, the probes within one probe spacing of the surface -- a shell
// whose volume per plane is proportional to the surface area crossing it, whatever its orientation -- and every rank holds
// the same 256 integers (same kernel, same scene: no communication).  The N - 1 cut planes put an N-th of that count into
// each slab, snapped to node boundaries of the sweep level (a node of that level must not straddle slabs) with at least
// one node per slab; with an empty histogram the equal slabs stay.
// Returns false (z_lo / z_hi untouched: the equal slabs) when there is nothing to balance on.
bool slab_cut_planes(const dcsg_ctx* c, const dcsg_export_config& cfg, float center_z, float diameter, int rank, int world, int& z_lo, int& z_hi)
{
    const int res = 1 << cfg.grid_level;
    const int sweep_level = cfg.min_level > 2 ? cfg.min_level : 2;
    const int level = sweep_level <= cfg.grid_level ? sweep_level : cfg.grid_level;
    const int node = res >> level;       // lattice planes per node of the sweep level
    const int n_nodes = 1 << level;      // candidate cut planes: node * k
    if (n_nodes < world || node < 1) return false;
    // work per node layer: the probe planes' counts spread over the lattice planes they cover
    std::vector<double> layer((size_t)n_nodes, 0.0);
    const double z0 = (double)center_z - 0.5 * (double)diameter, step = (double)diameter / res;
    double total = 0;
    for (int k = 0; k < 256; k++) {
        if (!c->probe_z_hist[k]) continue;
        const double z = (double)c->probe_half_neg + (double)c->probe_cd * (k - 128);
        double f = (z - z0) / step / node;  // node layer of the probe plane
        int l = (int)std::floor(f);
        l = l < 0 ? 0 : (l >= n_nodes ? n_nodes - 1 : l);
        layer[(size_t)l] += c->probe_z_hist[k];
        total += c->probe_z_hist[k];
    }
    if (!(total > 0)) return false;
    // every layer also carries a small constant (an empty slab still costs its launches): 2 % of the mean
    const double floor_w = 0.02 * total / n_nodes;
    for (auto& w : layer) w += floor_w;
    total += floor_w * n_nodes;
    std::vector<int> cut((size_t)world + 1, 0);
    cut[(size_t)world] = n_nodes;
    double acc = 0;
    int k = 1;
    for (int l = 0; l < n_nodes && k < world; l++) {
        acc += layer[(size_t)l];
        // the cut after layer l when the running sum passes k / world of the total; keep room for the slabs still to come
        while (k < world && acc >= total * k / world) {
            int at = l + 1;
            at = at < cut[(size_t)k - 1] + 1 ? cut[(size_t)k - 1] + 1 : at;
            at = at > n_nodes - (world - k) ? n_nodes - (world - k) : at;
            cut[(size_t)k++] = at;
        }
    }
    for (; k < world; k++) cut[(size_t)k] = std::max(cut[(size_t)k - 1] + 1, n_nodes - (world - k));
    z_lo = cut[(size_t)rank] * node;
    z_hi = cut[(size_t)rank + 1] * node;
    return true;
}

// Stages A .. projection of the export for Z slab `rank` of `world` (world = 1: the whole box).  Rank r owns the octree
// nodes whose lowest lattice plane lies in [r res/world, (r+1) res/world); coarse nodes that straddle
// slabs are classified by every rank from on-demand samples.  The lattice planes a slab's nodes read, including the
// plane it shares with the slab above, are evaluated by this rank itself (a sample is a pure function of its lattice
// index), so this entry point needs no exchange; the RCCL halo exchange of the dense-sweep path lives in
// designcsg_amd/sharded.py.  Triangles stay in the context's buffer.
int export_mesh_stage(dcsg_ctx* c, const dcsg_export_config& cfg, int rank, int world, unsigned flags, dcsg_export_stats& st,
                      bool project = true)
{
    int r;
    double t0 = now_s();
    // stage A (every rank: 0.2 ms, no communication)
    float diameter = 0;
    if ((r = dcsg_estimate_bounding_box(c, cfg.bounding_box_half_diameter, st.box_center, &diameter, &st.interior_points)))
        return r;
    st.box_diameter = diameter;
    st.t_bbox = now_s() - t0;
    if (!(diameter > 0)) return fail(c, DCSG_ERR_INVALID, "bounding-box estimate is empty");

    t0 = now_s();
    if (cfg.grid_level < 1 || cfg.grid_level > 12) return fail(c, DCSG_ERR_INVALID, "gridLevel out of range");
    const int res = 1 << cfg.grid_level;
    if (world < 1 || rank < 0 || rank >= world || res % world != 0)
        return fail(c, DCSG_ERR_INVALID, "the lattice does not split into that many equal Z slabs");
    int z_own_lo = rank * (res / world), z_own_hi = (rank + 1) * (res / world);
    if (world > 1 && !(flags & DCSG_EXPORT_EQUAL_SLABS)) slab_cut_planes(c, cfg, st.box_center[2], diameter, rank, world, z_own_lo, z_own_hi);
    st.slab_z_lo = z_own_lo;
    st.slab_z_hi = z_own_hi;
    const size_t n1 = (size_t)res + 1, plane = n1 * n1, n_pts = n1 * plane;
    const size_t slab_pts = (size_t)(z_own_hi - z_own_lo + 1) * plane;  // own planes + the shared upper one
    st.lattice_points = n_pts;
    // Normals: evaluated on demand by the mesher (the few nodes that reach the complex-edge test: a wave each,
    // dcsg_mesh_complex_edges), or swept densely with the SDF on request.  (Rounds 1-3 swept them densely for many-object
    // scenes, whose on-demand samples then visited every object: DCSG_DENSE_NORMALS=1 brings that back for comparison.)
    const bool dense_normals = (flags & DCSG_EXPORT_DENSE_NORMALS) != 0 ||
                               (c->variant == 4 && getenv("DCSG_DENSE_NORMALS") && !(flags & DCSG_EXPORT_SPARSE_NORMALS));
    // DCSG_DEBUG_LATTICE_GUARD=<bytes>: diagnostic.  The slab buffers get that many bytes of a known pattern in front and
    // behind, checked after the sweep and the mesher: a kernel that stores outside the slab's planes fails the export
    // instead of silently corrupting a neighbouring allocation (tests/test_gpu_fullsize.py runs the slab exports with it).
    const size_t guard = getenv("DCSG_DEBUG_LATTICE_GUARD") ? ((size_t)atol(getenv("DCSG_DEBUG_LATTICE_GUARD")) + 255) & ~(size_t)255 : 0;
    if ((r = ensure(c, c->d_lat_sdf, slab_pts * sizeof(float) + 2 * guard))) return r;
    if (dense_normals && (r = ensure(c, c->d_lat_nrm, slab_pts * 3 * sizeof(float) + 2 * guard))) return r;
    float* const lat_sdf = (float*)((char*)c->d_lat_sdf.p + guard);
    float* const lat_nrm = dense_normals ? (float*)((char*)c->d_lat_nrm.p + guard) : nullptr;
    if (guard) {
        HIPCHK(c, hipMemsetAsync(c->d_lat_sdf.p, 0xA5, guard, c->stream));
        HIPCHK(c, hipMemsetAsync((char*)lat_sdf + slab_pts * sizeof(float), 0xA5, guard, c->stream));
        if (dense_normals) {
            HIPCHK(c, hipMemsetAsync(c->d_lat_nrm.p, 0xA5, guard, c->stream));
            HIPCHK(c, hipMemsetAsync((char*)lat_nrm + slab_pts * 3 * sizeof(float), 0xA5, guard, c->stream));
        }
    }
    auto guards_intact = [&]() -> int {
        if (!guard) return 0;
        std::vector<unsigned char> h(guard);
        const void* at[4] = {c->d_lat_sdf.p, (char*)lat_sdf + slab_pts * sizeof(float), dense_normals ? c->d_lat_nrm.p : nullptr,
                             dense_normals ? (char*)lat_nrm + slab_pts * 3 * sizeof(float) : nullptr};
        for (int k = 0; k < 4; k++) {
            if (!at[k]) continue;
            HIPCHK(c, hipMemcpy(h.data(), at[k], guard, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < guard; i++)
                if (h[i] != 0xA5)
                    return fail(c, DCSG_ERR_HIP, std::string("lattice guard overwritten (") + (k & 1 ? "behind" : "in front of") +
                                                     (k < 2 ? " the SDF slab" : " the normal slab") + ", byte " + std::to_string(i) + ")");
        }
        return 0;
    };
    // the kernels index the lattice by absolute plane: the slab buffer stands for planes z_own_lo .. z_own_hi
    float* sdf0 = (float*)((uintptr_t)lat_sdf - (uintptr_t)z_own_lo * plane * sizeof(float));
    float* nrm0 = dense_normals ? (float*)((uintptr_t)lat_nrm - (uintptr_t)z_own_lo * plane * 3 * sizeof(float)) : nullptr;
    float diam3[3] = {diameter, diameter, diameter};
    // Node-wise sweep (default): the mesher's first levels sample on demand; at level max(min,2) -- the first whose
    // nodes the cull rule can drop for good -- the lattice is evaluated inside the nodes that are still alive, which
    // is everything later levels can read (a sample outside them, should rounding ever produce one, is evaluated on
    // demand: dcsg_lattice_point_written).  Worth it while a node is a small block; otherwise, or on request, the
    // whole slab is swept.  A node of the sweep level must not straddle slabs: 2^level >= world.
    const int sweep_level = cfg.min_level > 2 ? cfg.min_level : 2;
    const bool node_wise = !(flags & DCSG_EXPORT_DENSE_SWEEP) && !getenv("DCSG_DENSE_SWEEP") && sweep_level <= cfg.max_level &&
                           cfg.grid_level - sweep_level <= 4 && cfg.grid_level - sweep_level >= 0 && (1 << sweep_level) >= world;
    uint64_t n_tris = 0;
    if (node_wise) {
        NodeSweep ns;
        ns.level = sweep_level;
        ns.d_sdf = sdf0;
        ns.d_nrm = nrm0;
        if ((r = mesh_impl(c, st.box_center, diameter, &cfg, nullptr, nullptr, 0, res, z_own_lo, z_own_hi, &n_tris, &st.mesh, c->stream, &ns)))
            return r;
        st.sweep_kernel_ms = ns.ms;
        st.swept_nodes = ns.nodes;
        st.swept_points = ns.points;
        st.t_sweep = ns.ms * 1e-3;
        st.t_mesh = now_s() - t0 - st.t_sweep;
    } else {
        // one launch covers < 2^32 points; sweep in Z ranges when the slab is larger
        const int planes_per = (int)(0xf0000000ull / plane) > 0 ? (int)(0xf0000000ull / plane) : 1;
        double sweep_ms = 0;
        for (int z0 = z_own_lo; z0 <= z_own_hi; z0 += planes_per) {
            int z1 = z0 + planes_per < z_own_hi + 1 ? z0 + planes_per : z_own_hi + 1;
            float* sd = sdf0 + (size_t)z0 * plane;
            float* nr = dense_normals ? nrm0 + (size_t)z0 * plane * 3 : nullptr;
            if ((r = sweep_device(c, st.box_center, diam3, res, z0, z1, sd, nr, c->stream, true))) return r;
            double ms = 0;
            if ((r = dcsg_last_sweep_ms(c, &ms))) return r;
            sweep_ms += ms;
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        st.sweep_kernel_ms = sweep_ms;
        st.swept_points = slab_pts;
        st.t_sweep = now_s() - t0;

        // mesher (retopologize = identity: the reference's is undefined behaviour)
        t0 = now_s();
        if ((r = dcsg_mesh_lattice_device(c, st.box_center, diameter, &cfg, lat_sdf, lat_nrm, z_own_lo, z_own_hi, z_own_lo,
                                          z_own_hi, &n_tris, &st.mesh, c->stream)))
            return r;
        st.t_mesh = now_s() - t0;
    }
    if ((r = guards_intact())) return r;
    if (flags & DCSG_EXPORT_RETOPO) {  // opt-in; the default reproduces the reference's (repaired) identity
        // the repair looks up the mesh vertex occupying a cell: a slab only sees its own vertices, so T-junctions on slab
        // borders would stay open and the result would differ from the single-GPU mesh
        if (world > 1) return fail(c, DCSG_ERR_INVALID, "DCSG_EXPORT_RETOPO is not available for Z-slab exports (world > 1)");
        if ((r = dcsg_retopologize(c, st.box_center, diameter, cfg.min_level, cfg.grid_level, c->stream))) return r;
        n_tris = c->n_tris;
    }
    st.triangles = n_tris;
    if (!project) return 0;

    t0 = now_s();
    if ((r = dcsg_project_triangles(c, cfg.gradient_descent_steps, c->stream))) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    st.t_project = now_s() - t0;
    return 0;
}

// triangles of the context -> pinned host staging buffer (kept across exports: PCIe rate instead of pageable rate)
int stage_triangles_on_host(dcsg_ctx* c, float** host)
{
    size_t need = (size_t)c->n_tris * 9 * sizeof(float) + 64;
    if (need > c->h_pinned_cap) {
        if (c->h_pinned) hipHostFree(c->h_pinned);
        c->h_pinned = nullptr;
        c->h_pinned_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_pinned, need + need / 4, hipHostMallocDefault));
        c->h_pinned_cap = need + need / 4;
    }
    *host = (float*)c->h_pinned;
    return dcsg_download_triangles(c, *host, c->n_tris);
}


// The context's triangles -> stl_path / ply_path (reference layouts; either may be NULL) as triangles [first, first + n_tris)
// of files of `total`, streamed through the pinned ring: the copy engine fills a slot while the two writers (one thread per
// file, each holding its file open in a FileSink) drain earlier ones.  wflags: DCSG_WRITE_BEGIN is applied with the
// first piece, DCSG_WRITE_END with the last.  Tb / K: chunk boundaries of a pipelined projection -- a piece is downloaded
// once the chunk holding its last triangle is projected (chunk_ready[k], recorded on the context's stream); K = 0: the
// triangles are final (the caller has synchronised).  t_last / t_first: when the last / first piece was on the host.
int stream_triangles_to_files(dcsg_ctx* c, const char* stl_path, const char* ply_path, uint64_t first, uint64_t total, unsigned wflags,
                              int K, const uint64_t* Tb, double& dt_stl, double& dt_ply, double& t_last, double* t_first = nullptr)
{
    const uint64_t T = c->n_tris;
    const int n_writers = (stl_path ? 1 : 0) + (ply_path ? 1 : 0);
    dt_stl = dt_ply = t_last = 0;
    if (!n_writers) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ring_job) {  // what dcsg_create's thread pinned in the meantime
        if (c->ring_job->th.joinable()) c->ring_job->th.join();
        if (c->ring_job->p && !c->h_ring) {
            c->h_ring = c->ring_job->p;
            c->ring_slot_bytes = c->ring_job->slot_bytes;
        } else if (c->ring_job->p) {
            hipHostFree(c->ring_job->p);
        }
        c->ring_job.reset();
    }
    // a piece: a quarter of the mesh, between 1 and 16 MiB; the ring: four slots that hold a piece each (without the
    // pre-allocation a small mesh does not pay for pinning 64 MiB), grown when a larger mesh comes
    size_t want = (((size_t)T * 9 * sizeof(float) / dcsg_ctx::kRingSlots) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    want = want > dcsg_ctx::kRingSlotBytes ? dcsg_ctx::kRingSlotBytes : want;
    {
        if (!c->h_ring || want > c->ring_slot_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->copy_stream));
            if (c->h_ring) HIPCHK(c, hipHostFree(c->h_ring));
            c->h_ring = nullptr;
            HIPCHK(c, hipHostMalloc((void**)&c->h_ring, dcsg_ctx::kRingSlots * want, hipHostMallocDefault));
            c->ring_slot_bytes = want;
        }
        for (auto& e : c->ring_ev)
            if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int S = dcsg_ctx::kRingSlots;
    const size_t slot_bytes = c->ring_slot_bytes;
    const uint64_t P = (want / (9 * sizeof(float))) & ~(uint64_t)255;  // triangles per piece (want <= slot_bytes)
    // pieces never straddle the chunks of a pipelined projection (a piece is downloaded when ITS chunk is ready)
    struct Piece {
        uint64_t t, n;
        int chunk;
    };
    std::vector<Piece> pieces;
    for (int k = 0; k < (K > 0 ? K : 1); k++) {
        const uint64_t lo = K > 0 ? Tb[k] : 0, hi = K > 0 ? Tb[k + 1] : T;
        for (uint64_t t = lo; t < hi; t += P) pieces.push_back({t, std::min<uint64_t>(P, hi - t), k});
    }
    if (pieces.empty()) pieces.push_back({0, 0, 0});  // an empty mesh still carries BEGIN / END
    const uint64_t n_pieces = pieces.size();
    const float* d_tris = (const float*)c->d_tris.p;
    std::mutex m;
    std::condition_variable cv;
    uint64_t enqueued = 0;
    std::vector<int> finished((size_t)n_pieces, 0);
    bool failed = false;
    int rc_w[2] = {0, 0};
    double busy[2] = {0, 0};
    auto writer = [&](int idx, const char* path, bool ply) {
        const double a = now_s();
        hipSetDevice(c->device);
        FileSink sink;
        rc_w[idx] = sink.open(path, ply, first, T, total, (wflags & DCSG_WRITE_BEGIN) != 0);
        if (!rc_w[idx]) sink.write_faces_ahead(T);  // while the first piece is on its way
        for (uint64_t j = 0; j < n_pieces; j++) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return enqueued > j || failed; });
                if (failed) break;
            }
            const uint64_t t_j = pieces[(size_t)j].t, n_j = pieces[(size_t)j].n;
            if (!rc_w[idx] && n_j && hipEventSynchronize(c->ring_ev[j % S]) != hipSuccess) rc_w[idx] = DCSG_ERR_HIP;
            if (j == 0 && t_first && (!ply || !stl_path)) *t_first = now_s();
            if (j + 1 == n_pieces && (!ply || !stl_path)) t_last = now_s();
            if (!rc_w[idx]) sink.write((const float*)(c->h_ring + (j % S) * slot_bytes), t_j, n_j);
            if (!rc_w[idx] && j + 1 == n_pieces) rc_w[idx] = sink.close((wflags & DCSG_WRITE_END) != 0);
            {
                std::lock_guard<std::mutex> lk(m);
                finished[(size_t)j]++;
                if (rc_w[idx]) failed = true;
            }
            cv.notify_all();
            if (rc_w[idx]) break;
        }
        busy[idx] = now_s() - a;   // a sink still open here (failure) empties its file when it goes out of scope
    };
    std::thread th_stl, th_ply;
    if (stl_path) th_stl = std::thread([&] { writer(0, stl_path, false); });
    if (ply_path) th_ply = std::thread([&] { writer(1, ply_path, true); });
    // the director: slot j % S is free once both writers are done with piece j - S
    hipError_t he = hipSuccess;
    for (uint64_t j = 0; j < n_pieces && he == hipSuccess; j++) {
        if (j >= (uint64_t)S) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished[(size_t)(j - S)] == n_writers || failed; });
            if (failed) break;
        }
        const uint64_t t_j = pieces[(size_t)j].t, n_j = pieces[(size_t)j].n;
        if (n_j) {
            if (K > 0) he = hipStreamWaitEvent(c->copy_stream, c->chunk_ready[pieces[(size_t)j].chunk], 0);
            if (he == hipSuccess)
                he = hipMemcpyAsync(c->h_ring + (j % S) * slot_bytes, d_tris + t_j * 9, (size_t)n_j * 9 * sizeof(float), hipMemcpyDeviceToHost,
                                    c->copy_stream);
            if (he == hipSuccess) he = hipEventRecord(c->ring_ev[j % S], c->copy_stream);
        }
        {
            std::lock_guard<std::mutex> lk(m);
            if (he == hipSuccess) enqueued = j + 1;
            else failed = true;
        }
        cv.notify_all();
    }
    if (th_stl.joinable()) th_stl.join();
    if (th_ply.joinable()) th_ply.join();
    hipStreamSynchronize(c->copy_stream);
    dt_stl = busy[0];
    dt_ply = busy[1];
    if (he != hipSuccess) return fail(c, DCSG_ERR_HIP, std::string("triangle download: ") + hipGetErrorString(he));
    if (rc_w[0]) return fail(c, rc_w[0], "cannot write STL");
    if (rc_w[1]) return fail(c, rc_w[1], "cannot write PLY");
    return 0;
}

}  // namespace

extern "C" {

// Projection, download and both writers of dcsg_export as a pipeline over K chunks of the triangle buffer: while the
// host formats and writes chunk k, the GPU projects chunk k+1 and the copy engine downloads it.  Exact by
// construction: projection is a pure function of a vertex's position, so the order in which vertices are projected
// cannot change a bit; chunks are cut at triangle counts that are multiples of 256, where the scanned owner counts of
// the de-duplication give the compact-list boundary directly (every triangle before the cut refers only to owners
// before it: owners are the first occurrences, in buffer order).
int project_download_write(dcsg_ctx* c, int steps, const char* stl_path, const char* ply_path, unsigned flags, dcsg_export_stats& st)
{
    const uint64_t T = c->n_tris;
    const bool reference_writers = !(flags & (DCSG_EXPORT_INDEXED_PLY | DCSG_EXPORT_STL_NORMALS));
    const char* env = getenv("DCSG_EXPORT_CHUNKS");
    int K = env ? atoi(env) : 4;
    if (K > 8) K = 8;
    // Not for the tile-culling kernels: their projection is bound by the 50 dependent steps of a wave, not by throughput,
    // so a launch over a quarter of the vertices takes almost as long as one over all of them (flat256 1024^3: 41 ms in
    // 4 chunks against 15 ms in one).
    // ... and not for programs of many brush evaluations, for the same reason (userbrush128, 128 evaluations, 512^3: 33 ms in
    // four chunks, 24 in two, 21 in one; Design2 1024^3: 15.9 / 14.9 / 18.4)
    if (K < 2 || T < 200000 || !reference_writers || !(stl_path || ply_path) || T * 3 > (1ull << 30) || c->variant == 4 ||
        (c->n_object_evals >= 32 && !env))
        return 1;
    int r = need_scene(c);
    if (r) return r;
    const double t0 = now_s();
    uint64_t Tb[9];
    for (int k = 0; k <= K; k++) Tb[k] = k == K ? T : ((T * (uint64_t)k / (uint64_t)K) & ~(uint64_t)255);
    const bool dedup = dedup_applies(c, steps);
    unsigned Sb[9] = {0};
    Dedup d;
    if (dedup) {
        unsigned probe[8], got[8];
        for (int k = 1; k < K; k++) probe[k - 1] = (unsigned)(3 * Tb[k] / DCSG_BLOCK);
        if ((r = dedup_prepare(c, c->stream, d, probe, got, K - 1))) return r;
        for (int k = 1; k < K; k++) Sb[k] = got[k - 1];
        Sb[K] = d.n_unique;
    }
    float* d_tris = (float*)c->d_tris.p;
    for (int k = 0; k < K; k++) {
        const uint64_t t_lo = Tb[k], t_n = Tb[k + 1] - Tb[k];
        if (steps > 0) {
            if (dedup) {
                if ((r = dcsg_project_vertices_device(c, (float*)c->d_dd_compact.p + (size_t)Sb[k] * 3, Sb[k + 1] - Sb[k], steps, c->stream))) return r;
                if ((r = dedup_scatter(c, c->stream, (unsigned)(3 * t_lo), (unsigned)(3 * t_n)))) return r;
            } else if ((r = dcsg_project_vertices_device(c, d_tris + t_lo * 9, (size_t)t_n * 3, steps, c->stream))) {
                return r;
            }
        }
        HIPCHK(c, hipEventRecord(c->chunk_ready[k], c->stream));
    }
    double dt_stl = 0, dt_ply = 0, t_last = 0;
    r = stream_triangles_to_files(c, stl_path, ply_path, 0, T, DCSG_WRITE_BEGIN | DCSG_WRITE_END, K, Tb, dt_stl, dt_ply, t_last);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (r) return r;
    // the stages overlap: t_project = until the last piece was on the host (projection + its share of the downloads),
    // the writers' times include their waiting for pieces
    st.t_project = (t_last ? t_last : now_s()) - t0;
    st.t_download = 0;
    st.t_write_stl = dt_stl;
    st.t_write_ply = dt_ply;
    return 0;
}

// One Z slab of the export, stages A .. projection (no files): what each rank of a sharded export runs.
int dcsg_export_slab(dcsg_ctx* c, const char* cfg_txt, int rank, int world, unsigned flags, dcsg_export_stats* stats)
{
    int r = need_scene(c);
    if (r) return r;
    dcsg_export_config cfg;
    std::string err;
    if (!parse_export_config(cfg_txt, cfg, err)) return fail(c, DCSG_ERR_INVALID, err);
    dcsg_export_stats st;
    memset(&st, 0, sizeof(st));
    double t_start = now_s();
    if ((r = export_mesh_stage(c, cfg, rank, world, flags, st))) return r;
    st.t_total = now_s() - t_start;
    if (stats) *stats = st;
    return 0;
}

// The slab's triangles into the files at their final offsets (dcsg_write_*_range): triangles [first, first + own) of
// `total`.  flags: DCSG_WRITE_BEGIN / DCSG_WRITE_END as for the range writers (rank 0, around barriers).
int dcsg_write_slab(dcsg_ctx* c, const char* stl_path, const char* ply_path, uint64_t first, uint64_t total, unsigned flags)
{
    if (!c) return DCSG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the slab's projection
    double dt_stl = 0, dt_ply = 0, t_last = 0;
    return stream_triangles_to_files(c, stl_path, ply_path, first, total, flags, 0, nullptr, dt_stl, dt_ply, t_last);
}

// The sharded export inside ONE process: context i (one per GPU, each with the scene loaded) runs Z slab i of n on its
// own host thread, then every thread writes its byte range of the files.  What a C / C++ host that owns several GPUs
// calls instead of dcsg_export (the reference's own host is single-process: DesignCSG.cpp:638-790); the
// one-process-per-GPU form of the same thing is designcsg_amd/sharded.py: sharded_export_files.
int dcsg_export_multi(dcsg_ctx* const* ctxs, int n, const char* cfg_txt, const char* stl_path, const char* ply_path, unsigned flags,
                      dcsg_export_stats* stats)
{
    if (!ctxs || n < 1 || !cfg_txt) return DCSG_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return DCSG_ERR_INVALID;
    dcsg_export_config cfg;
    std::string err;
    if (!parse_export_config(cfg_txt, cfg, err)) return fail(ctxs[0], DCSG_ERR_INVALID, err);
    const double t_start = now_s();
    std::vector<dcsg_export_stats> st(n);
    std::vector<int> rc(n, 0);
    auto on_all = [&](auto&& f) {
        std::vector<std::thread> th;
        for (int i = 1; i < n; i++) th.emplace_back([&, i] { rc[i] = f(i); });
        rc[0] = f(0);
        for (auto& t : th) t.join();
        for (int i = 0; i < n; i++)
            if (rc[i]) return rc[i];
        return 0;
    };
    int r = on_all([&](int i) {
        int q = need_scene(ctxs[i]);
        memset(&st[i], 0, sizeof(st[i]));
        return q ? q : export_mesh_stage(ctxs[i], cfg, i, n, flags, st[i]);
    });
    if (r) {
        for (int i = 0; i < n; i++)
            if (rc[i] && i) ctxs[0]->err = "slab " + std::to_string(i) + ": " + ctxs[i]->err;
        return r;
    }
    uint64_t total = 0;
    std::vector<uint64_t> first(n);
    for (int i = 0; i < n; i++) {
        first[i] = total;
        total += st[i].triangles;
    }
    if (stl_path || ply_path) {
        if (stl_path && (r = dcsg_write_stl_range(stl_path, nullptr, 0, 0, total, DCSG_WRITE_BEGIN))) return fail(ctxs[0], r, "cannot write STL");
        if (ply_path && (r = dcsg_write_ply_range(ply_path, nullptr, 0, 0, total, DCSG_WRITE_BEGIN))) return fail(ctxs[0], r, "cannot write PLY");
        const double t0 = now_s();
        r = on_all([&](int i) { return dcsg_write_slab(ctxs[i], stl_path, ply_path, first[i], total, 0); });
        if (r) return r;
        if (stl_path && (r = dcsg_write_stl_range(stl_path, nullptr, 0, 0, total, DCSG_WRITE_END))) return fail(ctxs[0], r, "cannot write STL");
        if (ply_path && (r = dcsg_write_ply_range(ply_path, nullptr, 0, 0, total, DCSG_WRITE_END))) return fail(ctxs[0], r, "cannot write PLY");
        for (int i = 0; i < n; i++) st[i].t_write_stl = st[i].t_write_ply = now_s() - t0;
    }
    for (int i = 0; i < n; i++) st[i].t_total = now_s() - t_start;
    if (stats) memcpy(stats, st.data(), sizeof(dcsg_export_stats) * (size_t)n);
    return 0;
}

// MyFrame::OnExportInner, DesignCSG.cpp:638-790
int dcsg_export(dcsg_ctx* c, const char* cfg_txt, const char* stl_path, const char* ply_path, unsigned flags,
                dcsg_export_stats* stats)
{
    int r = need_scene(c);
    if (r) return r;
    dcsg_export_config cfg;
    std::string err;
    if (!parse_export_config(cfg_txt, cfg, err)) return fail(c, DCSG_ERR_INVALID, err);
    dcsg_export_stats st;
    memset(&st, 0, sizeof(st));
    double t_start = now_s(), t0;
    if ((r = export_mesh_stage(c, cfg, 0, 1, flags, st, false))) return r;
    const uint64_t n_tris = st.triangles;

    // projection + download + writers: pipelined over chunks where that applies (large meshes, reference-layout files) ...
    r = getenv("DCSG_NO_PIPELINE") ? 1 : project_download_write(c, cfg.gradient_descent_steps, stl_path, ply_path, flags, st);
    if (r < 0) return r;
    if (r == 0) {
        st.t_total = now_s() - t_start;
        if (stats) *stats = st;
        return 0;
    }
    // ... otherwise one after the other
    t0 = now_s();
    if ((r = dcsg_project_triangles(c, cfg.gradient_descent_steps, c->stream))) return r;
    if ((stl_path || ply_path) && !(flags & (DCSG_EXPORT_INDEXED_PLY | DCSG_EXPORT_STL_NORMALS))) {
        // the writers open their files (and the .ply gets its face records) while the projection runs: one "chunk", ready
        // when the kernel is
        HIPCHK(c, hipEventRecord(c->chunk_ready[0], c->stream));
        const uint64_t whole[2] = {0, n_tris};
        double dt_stl = 0, dt_ply = 0, t_last = 0, t_first = 0;
        r = stream_triangles_to_files(c, stl_path, ply_path, 0, n_tris, DCSG_WRITE_BEGIN | DCSG_WRITE_END, 1, whole, dt_stl, dt_ply, t_last, &t_first);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (r) return r;
        const double t1 = now_s();
        st.t_project = (t_first ? t_first : t1) - t0;  // until the first piece was on the host (the projection, then one download)
        st.t_download = t_last && t_first ? t_last - t_first : 0;  // the other pieces; the writers overlap them
        st.t_write_stl = dt_stl;  // both include their wait for the projection
        st.t_write_ply = dt_ply;
        st.t_total = now_s() - t_start;
        if (stats) *stats = st;
        return 0;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    st.t_project = now_s() - t0;
    if (stl_path || ply_path) {  // the opt-in writers need the whole mesh at once (vertex sharing, facet normals)
        t0 = now_s();
        float* host = nullptr;
        if ((r = stage_triangles_on_host(c, &host))) return r;
        st.t_download = now_s() - t0;
        t0 = now_s();
        int rc_stl = 0, rc_ply = 0;
        double dt_stl = 0, dt_ply = 0;
        std::thread ply_thread;
        if (ply_path)
            ply_thread = std::thread([&] {
                double a = now_s();
                rc_ply = (flags & DCSG_EXPORT_INDEXED_PLY) ? dcsg_write_ply_indexed(ply_path, host, n_tris, nullptr)
                                                           : dcsg_write_ply(ply_path, host, n_tris);
                dt_ply = now_s() - a;
            });
        if (stl_path) {
            double a = now_s();
            rc_stl = (flags & DCSG_EXPORT_STL_NORMALS) ? dcsg_write_stl_normals(stl_path, host, n_tris)
                                                       : dcsg_write_stl(stl_path, host, n_tris);
            dt_stl = now_s() - a;
        }
        if (ply_thread.joinable()) ply_thread.join();
        st.t_write_stl = dt_stl;  // the two writers overlap: their sum exceeds the wall time of this stage
        st.t_write_ply = dt_ply;
        if (rc_stl) return fail(c, rc_stl, "cannot write STL");
        if (rc_ply) return fail(c, rc_ply, "cannot write PLY");
    }
    st.t_total = now_s() - t_start;
    if (stats) *stats = st;
    return 0;
}

}  // extern "C"
