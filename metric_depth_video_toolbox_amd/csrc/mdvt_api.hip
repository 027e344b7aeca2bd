// mdvt_api.hip -- the C ABI of include/mdvt.h: context, parameter preparation, launch sequencing.
// Host code only; the kernels are in mdvt_kernels.hip.  Compiled with -ffp-contract=off (the f64
// composition of the eye matrices below is part of the arithmetic decree).
#include "mdvt_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <array>
#include <map>
#include <vector>

using namespace mdvt;
using namespace mdvt::grid8;          // grid-independent launchers of the rasterising translation units; the renders are dispatched:
// The render launchers exist once per sub-pixel grid (mdvt_internal.h); a context uses the set of its mdvt_config.subpixel_bits.
#define MDVT_GRID_CALL(c, fn, ...) (grid_bits(c) == 4 ? mdvt::grid4::fn(__VA_ARGS__) : mdvt::grid8::fn(__VA_ARGS__))

namespace {

thread_local std::string g_create_error;

constexpr int kParamSlots = 8;        // pinned staging ring for per-frame constants
constexpr int kWorkspaceChunk = 8;    // frames per launch when a global workspace is needed

struct ParamSlot {
    FrameDev* host = nullptr;         // pinned
    FrameDev* dev = nullptr;
    size_t capacity = 0;              // frames
    hipEvent_t done = nullptr;        // H2D copy + the kernels reading it have been submitted/finished
    bool used = false;
};

}  // namespace

struct mdvt_ctx {
    int device = 0;
    int pool_tag = 0;                 // the GPU whose pooled workspace blocks this context may take (= device; tuning build: MDVT_POOL_TAG)
    int W = 0, H = 0;
    mdvt_config cfg{};
    bool cfg_set = false;
    std::string err;
    ParamSlot slots[kParamSlots];
    int next_slot = 0;
    // the most recently staged parameter block: clips with constant parameters re-use the device copy
    std::vector<FrameDev> last_staged;
    ParamSlot* last_slot = nullptr;
    hipStream_t last_stream = nullptr;
    // workspace for the general path / edge filter, sized for ws_frames frames
    int ws_frames = 0;
    bool ws_keys = false, ws_ekeys = false, ws_edges = false;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned long long* ekeys[2] = {nullptr, nullptr};
    uint32_t* elist = nullptr;        // written edge-key words per (slot, source row) + counters (behind the entries)
    unsigned long long* cbuf[2] = {nullptr, nullptr};
    bool ws_mesh = false;
    bool keys_dirty = false;          // a general-path submission was interrupted between splat and resolve
    uint32_t key_parity = 0;          // bit s: parity of the next use of z-key slot s (mdvt_device.h, parity scheme)
    uint8_t* tri_invalid = nullptr;
    uint8_t* unused = nullptr;
    uint32_t* bigq = nullptr;         // general mesh path: queue of large triangles + its counter (last dword)
    uint32_t bigq_cap = 0;
    size_t bigq_bytes = 0;            // the queue block as laid out (without tuning padding)
    int huge_lists = 1;               // huge lists inside the queue block (2: tuning layout "joint")
    mdvt::RowCell* rowcell = nullptr; // [H] scanline -> cell row table of the mesh grid (pure-shift band kernel)
    int rowcell_bits = 0;             // the sub-pixel grid that table was built for
    uint32_t* row_counts = nullptr;   // [row_counts_frames][2][H]
    uint32_t* wave_counts = nullptr;  // [row_counts_frames][H][16] (RenderArgs.wave_counts)
    uint32_t* divcheck = nullptr;     // [kDivSlots] (RenderArgs.divcheck), zeroed when allocated; slot k belongs to div_keys[k]
    std::vector<std::array<uint32_t, 3>> div_keys;      // bits of (mult, scale, dl) of the parameter sets checked so far
    int row_counts_frames = 0;
    // infill-mask completion: per image stamp u16 + T f32 + work image u8x3, and the per-image counters
    int telea_images = 0, telea_rounds = 0;
    mdvt::TeleaWorkspace telea{};
    uint32_t* telea_levels_host = nullptr;      // pinned: the deepest level of a pass, read back once per pass
    // normal_infill / infill_using_mask_normals: about 16 B/px per image in flight
    uint8_t* ni_ws = nullptr;
    int ni_images = 0;
    // edge_row_range() of the most recent camera matrix (a clip's frames mostly share it)
    double erow_key[5] = {0, 0, 0, 0, 0};
    int erow_val[3] = {0, 0, 0};
    bool erow_cached = false;
    // every device allocation the context owns, by size (mdvt_workspace_bytes)
    std::unordered_map<void*, size_t> allocs;
    size_t ws_bytes = 0;
    bool opt_mesh_conv = false;       // MDVT_MESH_CONV=1 in the environment of mdvt_create (the opt-in kernel of mdvt_mesh_conv.hip)
    // posed / converged mesh runs of more than one launch set: the sets alternate between the caller's stream and this one, each on
    // its own half of the workspace slots (mdvt_render_stereo_batch); made on first use
    hipStream_t side = nullptr;
    uint32_t* hugeq2 = nullptr;
    hipEvent_t ev_start = nullptr, ev_join = nullptr, ev_vert[2] = {nullptr, nullptr};
};

namespace {

int fail(mdvt_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define MDVT_HIP(c, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail((c), MDVT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- Device workspace: a process-wide pool ---------------------------------------------------------------------------------
// A context's workspace blocks are NOT returned to the driver when the context goes: they wait here for the next context (of
// the same GPU) that asks for the same size class.  Why: the r04 soak found one FIRST render of a fresh context in ~3 000
// (12 processes sharing the GPU, a context created and destroyed per render) that lost entries of the triangle queue, and
// one process killed by a GPU memory fault -- only when the queue's block was larger than 2 MB, i.e. when it no longer came
// out of the runtime's own cache of sub-2 MB fragments but was mapped by hipMalloc and unmapped by hipFree once per context;
// never on a context's later renders, never with HSA_ENABLE_SDMA=0.  DESIGN.md section 9 has the diagnosis (r05: what the
// lost words held, which treatments of a fresh block stop it; tools/probe/fresh_alloc_probe.hip is the pattern without the
// library).  Whatever the cause below the HIP API, the library no longer creates the condition: (1) a block that does come
// fresh from hipMalloc is filled and the stream synchronised before anything uses it, (2) blocks are recycled here instead of
// freed, so in steady state no render ever runs on memory that was mapped microseconds earlier, (3) idle blocks are only given
// back to the driver beyond kDevPoolIdleCap bytes (oldest first) or on mdvt_release_cached_memory, each time behind a
// hipDeviceSynchronize.  A recycled block holds a previous user's data: nothing in the library reads a workspace word before
// the same call has written it (the soaks' sub-2 MB blocks always were recycled this way, by the runtime).
// The same treatment the pinned parameter blocks got in r03 (pool_take / pool_give below).
// Tuning build: MDVT_WS_POOL=off -> hipMalloc / hipFree per context as until r04; MDVT_WS_FRESH=none|canary|devsync|memset picks
// the treatment of a fresh block (product: memset); MDVT_POOL_TAG=n labels this context's blocks as GPU n's (tests).
struct DevBlock { void* p; size_t bytes; int tag; unsigned long long stamp; };
std::mutex g_dev_pool_mutex;
std::vector<DevBlock>& dev_pool() { static std::vector<DevBlock> p; return p; }
std::map<int, size_t>& dev_pool_idle() { static std::map<int, size_t> m; return m; }      // idle bytes per pool tag (= per GPU)
unsigned long long g_dev_pool_stamp = 0;
// Idle bytes kept PER GPU before that GPU's oldest blocks go back to the driver (mdvt_set_cached_memory_limit; default 4 GiB = the
// default workspace_mib budget, i.e. one context's worth of the largest workspace the library allocates by default).
size_t g_dev_pool_idle_cap = (size_t)4 << 30;

// Size classes: 4 KiB steps up to 64 KiB, 16 steps per power of two up to 1 MiB (at most 6.25 % over the request), 64 KiB steps
// above (the large blocks are what mdvt_config.workspace_mib budgets: they stay what was asked for; a clip's contexts share
// one frame size, so their blocks match exactly anyway).
size_t ws_size_class(size_t bytes)
{
    if (bytes <= ((size_t)64 << 10)) return (bytes + 4095) & ~(size_t)4095;
    if (bytes > ((size_t)1 << 20)) return (bytes + 65535) & ~(size_t)65535;
    size_t step = (size_t)4096;
    while ((step << 5) < bytes) step <<= 1;              // bytes in (16 step, 32 step]
    return (bytes + step - 1) / step * step;
}
bool dev_pool_off() { const char* e = tuning_env(TUNE_WS_POOL); return e && (strcmp(e, "off") == 0 || strcmp(e, "delay") == 0); }

// device memory owned by a context, accounted for mdvt_workspace_bytes; `s`: the stream the fresh-block fill goes to
hipError_t ws_malloc(mdvt_ctx* c, void** p, size_t bytes, hipStream_t s)
{
    *p = nullptr;
    const bool pooled = !dev_pool_off();
    const size_t want = pooled ? ws_size_class(bytes) : bytes;
    if (pooled) {
        std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
        auto& pool = dev_pool();
        for (size_t k = pool.size(); k-- > 0;)            // newest first
            if (pool[k].bytes == want && pool[k].tag == c->pool_tag) {
                *p = pool[k].p;
                dev_pool_idle()[c->pool_tag] -= want;
                pool.erase(pool.begin() + (long)k);
                break;
            }
    }
    if (!*p) {
        const char* fresh0 = tuning_env(TUNE_WS_FRESH);
        hipError_t e;
        if (fresh0 && strcmp(fresh0, "uncached") == 0) e = hipExtMallocWithFlags(p, want, hipDeviceMallocUncached);          // (r05 diagnosis)
        else if (fresh0 && strcmp(fresh0, "finegrained") == 0) e = hipExtMallocWithFlags(p, want, hipDeviceMallocFinegrained);
        else e = hipMalloc(p, want);
        if (e != hipSuccess && pooled) {                  // out of memory with idle blocks of other classes around: give them back, once
            (void)hipGetLastError();
            mdvt_release_cached_memory(-1);
            e = hipMalloc(p, want);
        }
        if (e != hipSuccess) return e;
        const char* fresh = tuning_env(TUNE_WS_FRESH);
        if (!fresh || strcmp(fresh, "memset") == 0) {
            if ((e = hipMemsetAsync(*p, 0, want, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) { (void)hipFree(*p); *p = nullptr; return e; }
        } else if (strcmp(fresh, "canary") == 0) {
            if ((e = hipMemsetAsync(*p, 0xC5, want, s)) != hipSuccess) { (void)hipFree(*p); *p = nullptr; return e; }
        } else if (strcmp(fresh, "devsync") == 0) {
            if ((e = hipDeviceSynchronize()) != hipSuccess) { (void)hipFree(*p); *p = nullptr; return e; }
        }                                                 // "none": as until r04
    }
    c->allocs[*p] = want; c->ws_bytes += want;
    return hipSuccess;
}
// (the caller has made sure no submitted work still uses the block: mdvt_destroy and the growing paths synchronise the device)
void ws_free(mdvt_ctx* c, void* p)
{
    if (!p) return;
    size_t bytes = 0;
    auto it = c->allocs.find(p);
    if (it != c->allocs.end()) { bytes = it->second; c->ws_bytes -= bytes; c->allocs.erase(it); }
    if (dev_pool_off() || bytes == 0 || bytes != ws_size_class(bytes)) {
        // (r05 diagnosis, tuning build: MDVT_WS_POOL=delay -> a freed block waits behind the next 64 before it goes back to the driver,
        //  so its address range is not handed out again at once)
        const char* e = tuning_env(TUNE_WS_POOL);
        if (e && strcmp(e, "delay") == 0) {
            static std::vector<void*> ring;
            ring.push_back(p);
            if (ring.size() > 64) { (void)hipFree(ring.front()); ring.erase(ring.begin()); }
            return;
        }
        (void)hipFree(p);
        return;
    }
    std::vector<void*> out;
    {
        std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
        auto& pool = dev_pool();
        pool.push_back({p, bytes, c->pool_tag, ++g_dev_pool_stamp});
        size_t& idle = dev_pool_idle()[c->pool_tag];                                     // (accounted and capped per GPU)
        idle += bytes;
        for (size_t k = 0; idle > g_dev_pool_idle_cap && k < pool.size();) {             // oldest first (the vector is in stamp order)
            if (pool[k].tag != c->pool_tag) { ++k; continue; }                           // (this GPU's only: the device guard is the caller's)
            out.push_back(pool[k].p);
            idle -= pool[k].bytes;
            pool.erase(pool.begin() + (long)k);
        }
    }
    if (!out.empty()) {
        (void)hipDeviceSynchronize();
        for (void* q : out) (void)hipFree(q);
    }
}

// Everything the kernels need about one frame, derived in f64 and rounded once to f32.
// Pure-shift frames: on which row does the chain (mdvt_device.h "edge points") put an edge point of source row i?  Without
// pose and convergence the row is round( ((gy - cy) z / fy sH) (1/z) fyr + cyr ): in exact arithmetic independent of z,
//   v*(i) = (gy_i - cy) sH (fyr / fy) + cyr  ~  i + 1/2 - i / H^2   (mesh grid, cy = H/2),
// a hair below the tie i + 1/2 -- by less than the f32 rounding of fy (dmt:1058) moves it for the first rows, which then land
// on i + 1 -- and the eight f64 roundings of the chain move v by at most 8 H 2^-53.  Rows whose v* keeps a margin of four
// times that from a tie have their row decided here, once per camera matrix; the others (a tie in exact arithmetic: the
// roundings of each point decide) and the rows that land on i + 1 form [erow_lo, erow_hi), left to k_edge_rows_exact.
static void edge_row_range(FrameDev& f, int H)
{
    const long double fy = f.Kd[1], cy = f.Kd[3], sH = f.sHd, fyr = (long double)f.fyr, cyr = (long double)f.cyr;
    const long double margin = 32.0L * 1.1102230246251565e-16L * ((long double)H + fabsl(cyr) + 1.0L);
    int lo = H, hi = 0;
    bool wild = false;
    for (int i = 0; i < H; ++i) {
        const long double gy = f.sy == 1.0f ? (long double)i : (long double)((float)i * f.sy);
        const long double v = (gy - cy) * sH * (fyr / fy) + cyr;
        const long double fl = floorl(v);
        const bool undecided = fabsl(v - (fl + 0.5L)) <= margin;
        const long double row = undecided ? fl : floorl(v + 0.5L);           // undecided: fl or fl + 1
        const bool plain = !undecided && row == (long double)i;
        if (plain) continue;
        if (row != (long double)i && !(!undecided && row == (long double)i + 1.0L)) { wild = true; break; }
        if (i < lo) lo = i;
        if (i + 1 > hi) hi = i + 1;
    }
    f.erow_wild = wild ? 1 : 0;
    f.erow_lo = (wild || lo >= hi) ? 0 : lo;
    f.erow_hi = (wild || lo >= hi) ? 0 : hi;
}

int fill_frame_dev(mdvt_ctx* c, const mdvt_frame_params& p, FrameDev& f)
{
    const mdvt_config& cfg = c->cfg;
    const int W = c->W, H = c->H;
    const double fx = p.K[0], fy = p.K[4], cx = p.K[2], cy = p.K[5];
    const double fxr = p.Krender[0], fyr = p.Krender[4], cxr = p.Krender[2], cyr = p.Krender[5];
    if (!(fx > 0.0) || !(fy > 0.0) || !(fxr > 0.0) || !(fyr > 0.0))
        return fail(c, MDVT_ERR_INVALID_ARG, "camera matrix needs positive focal lengths");
    if (cxr * 2.0 != (double)W || cyr * 2.0 != (double)H)
        return fail(c, MDVT_ERR_UNSUPPORTED,
                    "render size (2*cx, 2*cy) = (%g, %g) differs from the frame size %dx%d (--vr180 is not built)",
                    cxr * 2.0, cyr * 2.0, W, H);
    if (!(p.depth_scale > 0.0)) return fail(c, MDVT_ERR_INVALID_ARG, "depth_scale must be > 0");
    memset(&f, 0, sizeof f);
    f.mult = (float)(cfg.max_depth / 4228250625.0);
    f.scale = (float)p.depth_scale;
    const double half = cfg.ipd_m / 2.0;
    f.dl = (float)(fxr * half);
    f.fx = (float)fx; f.fy = (float)fy; f.cx = (float)cx; f.cy = (float)cy;
    f.fxr = (float)fxr; f.fyr = (float)fyr; f.cxr = (float)cxr; f.cyr = (float)cyr;
    const bool mesh = cfg.mode == MDVT_MODE_MESH;
    f.sx = mesh ? (float)(((double)W + 1.0) / (double)W) : 1.0f;
    f.sy = mesh ? (float)(((double)H + 1.0) / (double)H) : 1.0f;
    f.sW = (float)(((double)W - 1.0) / (double)W);
    f.sH = (float)(((double)H - 1.0) / (double)H);
    f.Kd[0] = fx; f.Kd[1] = fy; f.Kd[2] = cx; f.Kd[3] = cy;
    f.rKd[0] = 1.0 / fx; f.rKd[1] = 1.0 / fy;
    const double conv = (p.convergence_angle == p.convergence_angle) ? p.convergence_angle : 0.0;   // NaN -> none
    const bool same_k = fx == fxr && fy == fyr && cx == cxr && cy == cyr;
    f.general = (p.has_T || conv != 0.0 || !same_k) ? 1 : 0;
    double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (p.has_T) {
        memcpy(T, p.T, sizeof T);
        if (fabs(T[12]) > 1e-12 || fabs(T[13]) > 1e-12 || fabs(T[14]) > 1e-12 || fabs(T[15] - 1.0) > 1e-12)
            return fail(c, MDVT_ERR_UNSUPPORTED, "pose matrix must be affine (last row 0 0 0 1)");
    }
    for (int eye = 0; eye < 2; ++eye) {
        // M = Translate(+-ipd/2) * Ry(-+a) * T;  Ry(t) = [[c,0,s],[0,1,0],[-s,0,c]]
        const double t = eye == 0 ? -conv : conv;
        const double cs = cos(t), sn = sin(t);
        const double R[3][3] = {{cs, 0.0, sn}, {0.0, 1.0, 0.0}, {-sn, 0.0, cs}};
        const double shift[3] = {eye == 0 ? half : -half, 0.0, 0.0};
        for (int r = 0; r < 3; ++r) {
            for (int col = 0; col < 3; ++col)
                f.M[eye][4 * r + col] = (float)((R[r][0] * T[0 + col] + R[r][1] * T[4 + col]) + R[r][2] * T[8 + col]);
            f.M[eye][4 * r + 3] = (float)(((R[r][0] * T[3] + R[r][1] * T[7]) + R[r][2] * T[11]) + shift[r]);
        }
    }
    // The edge points' chain takes the reference's operands as they are (mdvt_device.h "edge points")
    f.sWd = ((double)W - 1.0) / (double)W;
    f.sHd = ((double)H - 1.0) / (double)H;
    f.hd = half;
    f.has_T = p.has_T ? 1 : 0;
    memcpy(f.Td, T, sizeof T);
    f.has_conv = conv != 0.0 ? 1 : 0;
    f.cs[0] = cos(conv); f.cs[1] = sin(conv);
    if (!f.general && cfg.remove_edges && cfg.edge_points) {
        const double key[5] = {f.Kd[1], f.Kd[3], (double)f.fyr, (double)f.cyr, (double)f.sy};
        if (!(c->erow_cached && memcmp(key, c->erow_key, sizeof key) == 0)) {
            edge_row_range(f, H);
            memcpy(c->erow_key, key, sizeof key);
            c->erow_val[0] = f.erow_lo; c->erow_val[1] = f.erow_hi; c->erow_val[2] = f.erow_wild;
            c->erow_cached = true;
        }
        f.erow_lo = c->erow_val[0]; f.erow_hi = c->erow_val[1]; f.erow_wild = c->erow_val[2];
    }
    // Convergence and nothing else (sr:707-726: rotation about the camera's y axis, shift along x): the projected row of a
    // vertex is depth independent, v = (gy - cy) / rz(j) + cy with rz(j) = m10 + m8 (gx_j - cx) / fx, which k_mesh_conv
    // (mdvt_mesh_conv.hip) builds on.  It takes the frame if the vertex rows stay low staircases: at most 12 rows of tilt
    // across the frame (its row tags are 5 bits, its column pairs expect neighbouring brackets to differ by one).
    f.conv_band = 0;
    if (mesh && !p.has_T && conv != 0.0 && same_k) {
        bool ok = true;
        for (int eye = 0; eye < 2 && ok; ++eye) {
            const float* M = f.M[eye];
            ok = M[1] == 0.0f && M[4] == 0.0f && M[5] == 1.0f && M[6] == 0.0f && M[7] == 0.0f && M[9] == 0.0f && M[11] == 0.0f;
            const double rz0 = (double)M[10] + (double)M[8] * ((0.0 - cx) / fx);
            const double rz1 = (double)M[10] + (double)M[8] * (((double)(W - 1) * ((double)W + 1.0) / (double)W - cx) / fx);
            if (!(rz0 > 0.5 && rz1 > 0.5 && rz0 < 2.0 && rz1 < 2.0)) ok = false;
            else if (fabs(1.0 / rz0 - 1.0 / rz1) * ((double)H * 0.5 + 1.0) * (fyr / fy) > 12.0) ok = false;
        }
        f.conv_band = ok ? 1 : 0;
    }
    return MDVT_OK;
}

// Pinned host staging memory is NEVER returned to the driver while the process lives.  The r03 parity soak found one frame in
// ~20 000 context create / render / destroy cycles (14 processes sharing the GPU) rendered with the PREVIOUS context's
// parameter block: a hipHostMalloc'ed buffer that recycles the address of one just hipHostFree'd can be read by the GPU --
// copy engine or kernel alike, even with a stream synchronisation after the copy -- with the old allocation's content
// (tests/dbg_param_stress.py reproduces it: 12 wrong frames in 191 000 contexts with per-context hipHostMalloc /
// hipHostFree, 0 in 1 064 000 with this pool, and a context is created 5 x faster).  MDVT_PARAM_UPLOAD=recycle restores
// the per-context allocation for that A/B.
struct PoolBlock { void* host; void* dev; size_t bytes; int device; };     // device: the GPU `dev` was allocated on (-1: no device block)
std::mutex g_pool_mutex;
std::vector<PoolBlock>& param_pool() { static std::vector<PoolBlock> p; return p; }
bool param_pool_off()
{
    const char* e = tuning_env(TUNE_PARAM_UPLOAD);
    return e && strcmp(e, "recycle") == 0;
}
// A pinned host block of at least `bytes`, with a device block of the same size on GPU `device` if with_dev (a block that
// carries device memory only ever goes back to a context on the GPU it was allocated on: contexts of two GPUs share the pool).
hipError_t pool_take(size_t bytes, bool with_dev, int device, void** host, void** dev, size_t* got)
{
    if (!param_pool_off()) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto& pool = param_pool();
        for (size_t k = 0; k < pool.size(); ++k)
            if (pool[k].bytes >= bytes && pool[k].device == (with_dev ? device : -1)) {
                *host = pool[k].host; *dev = pool[k].dev; *got = pool[k].bytes;
                pool.erase(pool.begin() + (long)k);
                return hipSuccess;
            }
    }
    *host = nullptr; *dev = nullptr; *got = bytes;
    hipError_t e = hipHostMalloc(host, bytes, hipHostMallocDefault);
    if (e == hipSuccess && with_dev) {
        e = hipMalloc(dev, bytes);                       // (the caller's DeviceGuard has made `device` current)
        if (e != hipSuccess) { (void)hipHostFree(*host); *host = nullptr; *dev = nullptr; }
    }
    return e;
}
void pool_give(void* host, void* dev, size_t bytes, int device)
{
    if (!host) return;
    if (param_pool_off()) { (void)hipHostFree(host); if (dev) (void)hipFree(dev); return; }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    param_pool().push_back({host, dev, bytes, dev ? device : -1});
}

// Stage n FrameDev records to the device through the pinned ring; returns the device pointer.
int stage_params(mdvt_ctx* c, const std::vector<FrameDev>& v, hipStream_t s, const FrameDev** dev, ParamSlot** slot_out)
{
    if (c->last_slot && c->last_stream == s && c->last_staged.size() == v.size() &&
        memcmp(c->last_staged.data(), v.data(), v.size() * sizeof(FrameDev)) == 0) {
        // identical to what already sits on the device (stream order keeps the earlier copy ahead of us)
        *dev = c->last_slot->dev;
        *slot_out = c->last_slot;
        return MDVT_OK;
    }
    ParamSlot& sl = c->slots[c->next_slot];
    c->next_slot = (c->next_slot + 1) % kParamSlots;
    if (sl.used) MDVT_HIP(c, hipEventSynchronize(sl.done));     // slot is being reused: its last user must be done
    if (sl.capacity < v.size()) {
        pool_give(sl.host, sl.dev, sl.capacity * sizeof(FrameDev), c->pool_tag);
        sl.host = nullptr; sl.dev = nullptr; sl.capacity = 0;
        size_t cap = 16;
        while (cap < v.size()) cap *= 2;
        void *h = nullptr, *d = nullptr;
        size_t got = 0;
        MDVT_HIP(c, pool_take(cap * sizeof(FrameDev), true, c->pool_tag, &h, &d, &got));
        sl.host = (FrameDev*)h; sl.dev = (FrameDev*)d; sl.capacity = got / sizeof(FrameDev);
    }
    if (!sl.done) MDVT_HIP(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    memcpy(sl.host, v.data(), v.size() * sizeof(FrameDev));
    MDVT_HIP(c, hipMemcpyAsync(sl.dev, sl.host, v.size() * sizeof(FrameDev), hipMemcpyHostToDevice, s));
    sl.used = true;
    c->last_staged = v;
    c->last_slot = &sl;
    c->last_stream = s;
    *dev = sl.dev;
    *slot_out = &sl;
    return MDVT_OK;
}

// (the EMPTY fill of fresh key buffers goes on the caller's stream: PyTorch's pool streams do not synchronise with the
//  legacy null stream, so a fill issued there could land after the first splat)
int ensure_workspace(mdvt_ctx* c, int frames, bool need_keys, bool need_ekeys, bool need_edges, bool need_mesh_ws, hipStream_t s)
{
    const size_t npx = (size_t)c->W * c->H;
    const size_t ntri = 2 * (size_t)(c->W - 1) * (c->H - 1);
    const bool grow = frames > c->ws_frames;
    // (blocks that are replaced go back to the pool, where another context may pick them up at once: whatever was submitted
    //  with them -- to any stream -- has to be through first; hipFree used to wait for that implicitly)
    if (grow && c->ws_bytes) MDVT_HIP(c, hipDeviceSynchronize());
    // (not growing, yet a group that is not complete holds a buffer: an earlier call failed half-way through allocating it -- its
    //  asynchronous fill may still be pending, and the block must not reach the pool before that is through)
    if (!grow && ((need_keys && !c->ws_keys && (c->keys[0] || c->keys[1])) || (need_ekeys && !c->ws_ekeys && (c->ekeys[0] || c->ekeys[1] || c->elist)) ||
                  (need_edges && !c->ws_edges && (c->tri_invalid || c->unused)) || (need_mesh_ws && !c->ws_mesh && (c->cbuf[0] || c->cbuf[1]))))
        MDVT_HIP(c, hipDeviceSynchronize());
    if (grow || (need_keys && !c->ws_keys)) {
        for (int e = 0; e < 2; ++e) { if (c->keys[e]) ws_free(c, c->keys[e]); c->keys[e] = nullptr; }
        c->ws_keys = false;
    }
    if (grow || (need_ekeys && !c->ws_ekeys)) {
        for (int e = 0; e < 2; ++e) { if (c->ekeys[e]) ws_free(c, c->ekeys[e]); c->ekeys[e] = nullptr; }
        if (c->elist) ws_free(c, c->elist);
        c->elist = nullptr;
        c->ws_ekeys = false;
    }
    if (grow || (need_edges && !c->ws_edges)) {
        if (c->tri_invalid) ws_free(c, c->tri_invalid);
        if (c->unused) ws_free(c, c->unused);
        c->tri_invalid = nullptr; c->unused = nullptr; c->ws_edges = false;
    }
    if (grow || (need_mesh_ws && !c->ws_mesh)) {
        for (int e = 0; e < 2; ++e) { if (c->cbuf[e]) ws_free(c, c->cbuf[e]); c->cbuf[e] = nullptr; }
        c->ws_mesh = false;
    }
    if (grow) c->ws_frames = frames;
    const size_t nf = (size_t)c->ws_frames;
    if (need_keys && !c->ws_keys) {
        for (int e = 0; e < 2; ++e) {
            MDVT_HIP(c, ws_malloc(c, (void**)&c->keys[e], nf * npx * sizeof(unsigned long long), s));
            MDVT_HIP(c, hipMemsetAsync(c->keys[e], 0xFF, nf * npx * sizeof(unsigned long long), s));     // parity 0's empty value
        }
        c->key_parity = 0;
        c->ws_keys = true;
    }
    if (need_ekeys && !c->ws_ekeys) {
        for (int e = 0; e < 2; ++e) {
            MDVT_HIP(c, ws_malloc(c, (void**)&c->ekeys[e], nf * npx * sizeof(unsigned long long), s));
            MDVT_HIP(c, hipMemsetAsync(c->ekeys[e], 0xFF, nf * npx * sizeof(unsigned long long), s));
        }
        // (+ the vertex list of the mesh path's edge-point splat: npx entries and a counter per slot)
        MDVT_HIP(c, ws_malloc(c, (void**)&c->elist, (nf * 2 * npx + nf * (size_t)c->H + nf * npx + nf) * sizeof(uint32_t), s));
        MDVT_HIP(c, hipMemsetAsync(c->elist + nf * 2 * npx, 0, nf * (size_t)c->H * sizeof(uint32_t), s));   // counters; the reset pass keeps them 0
        c->ws_ekeys = true;
    }
    if (need_mesh_ws && !c->ws_mesh) {
        for (int e = 0; e < 2; ++e) {
            MDVT_HIP(c, ws_malloc(c, (void**)&c->cbuf[e], nf * npx * sizeof(unsigned long long), s));   // tie side words: a word is initialised by the fragment that marks its pixel, so the plane needs no clearing
        }
        if (c->bigq) ws_free(c, c->bigq);
        c->bigq = nullptr;
        // one segment per (frame slot, cell row), each with room for all four triangles of every cell of the row (8 bytes
        // per triangle) and its own counter: the queue cannot overflow
        // (entry indices are 32-bit: chunk_of() keeps nf * npx * 4 below 2^32, so allocation and counter offset use ONE value)
        size_t nfq = (size_t)0xFFFFFFF0u / (4 * npx);          // a general launch set never has more slots than this (chunk_of)
        if (nfq < 1) return fail(c, MDVT_ERR_UNSUPPORTED, "general mesh path: a %d x %d frame exceeds the 32-bit triangle queue", c->W, c->H);
        if (nfq > nf) nfq = nf;
        const size_t cap = nfq * npx * 4;
        c->bigq_cap = (uint32_t)cap;
        // entries, counters, prefix sums; row blocks of huge triangles + their counter; tie flags and tile bits
        // (tuning build, the r04 diagnosis: MDVT_WS_LAYOUT=joint puts the second bank's huge list back into this block, as at
        //  47b4117 -- 2.2 MB for a 100 x 31 frame; MDVT_WS_PAD=n appends n unused bytes)
        c->huge_lists = 1;
        if (const char* e = tuning_env(TUNE_WS_LAYOUT)) c->huge_lists = strcmp(e, "joint") == 0 ? 2 : 1;
        size_t pad = 0;
        if (const char* e = tuning_env(TUNE_WS_PAD)) pad = (size_t)strtoull(e, nullptr, 10);
        c->bigq_bytes = (cap * mdvt::kBigRecDwords + 2 * nf * (size_t)c->H + 8 + (size_t)c->huge_lists * (2 * (size_t)mdvt::kHugeCap + 2) + nf * (1 + 2 * mdvt::tie_words_of(c->W, c->H))) * sizeof(uint32_t);
        MDVT_HIP(c, ws_malloc(c, (void**)&c->bigq, c->bigq_bytes + pad, s));
        c->ws_mesh = true;
    }
    if (need_edges && !c->ws_edges) {
        MDVT_HIP(c, ws_malloc(c, (void**)&c->tri_invalid, nf * ntri, s));
        MDVT_HIP(c, ws_malloc(c, (void**)&c->unused, nf * npx, s));
        c->ws_edges = true;
    }
    return MDVT_OK;
}

bool aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// The decree's snap (mdvt_device.h) on the host: same IEEE operations (this file is compiled with -ffp-contract=off).
int host_snap(float x, int subpix)
{
    x = fminf(fmaxf(x, -kSnapLimit), kSnapLimit);
    return (int)rintf(x * (float)subpix);
}

int grid_bits(const mdvt_ctx* c) { return c->cfg.subpixel_bits == 4 ? 4 : 8; }

// Scanline k (centre S k + S/2) is covered by the cell row c = largest i with snap(f32(i) * sy) < centre (a centre ON a vertex
// row belongs to the cells above it: bottom edges own their centres, mdvt_device.h edge_in), if that is not the last vertex
// row (k_mesh_rows derives the same per workgroup).
int ensure_rowcell(mdvt_ctx* c, hipStream_t s)
{
    if (c->rowcell && c->rowcell_bits == grid_bits(c)) return MDVT_OK;
    const int H = c->H;
    const int kSubpix = 1 << grid_bits(c);        // (shadows the compile-time grid of this translation unit on purpose)
    const float sy = (float)(((double)H + 1.0) / (double)H);
    std::vector<mdvt::RowCell> t((size_t)H);
    for (int k = 0; k < H; ++k) {
        const int Yc = k * kSubpix + kSubpix / 2;
        int ilo = (int)(((float)k + 0.5f) / sy);
        ilo = ilo < 0 ? 0 : (ilo > H - 1 ? H - 1 : ilo);
        while (ilo > 0 && host_snap((float)ilo * sy, kSubpix) >= Yc) --ilo;
        while (ilo + 1 <= H - 1 && host_snap((float)(ilo + 1) * sy, kSubpix) < Yc) ++ilo;
        mdvt::RowCell r{};
        r.c = (ilo <= H - 2) ? ilo : -1;
        r.Yt = r.c >= 0 ? host_snap((float)r.c * sy, kSubpix) : 0;
        r.Yb = r.c >= 0 ? host_snap((float)(r.c + 1) * sy, kSubpix) : 1;
        t[(size_t)k] = r;
    }
    if (!c->rowcell) MDVT_HIP(c, ws_malloc(c, (void**)&c->rowcell, (size_t)H * sizeof(mdvt::RowCell), s));
    else MDVT_HIP(c, hipDeviceSynchronize());     // a render of the other grid may still read the table
    c->rowcell_bits = grid_bits(c);
    MDVT_HIP(c, hipMemcpyAsync(c->rowcell, t.data(), (size_t)H * sizeof(mdvt::RowCell), hipMemcpyHostToDevice, s));
    MDVT_HIP(c, hipStreamSynchronize(s));      // `t` is pageable host memory
    return MDVT_OK;
}

}  // namespace

extern "C" {

int mdvt_version(void) { return MDVT_VERSION; }

const char* mdvt_last_error(const mdvt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mdvt_create(mdvt_ctx** out, int device, int width, int height, uint32_t flags)
{
    if (!out) return fail(nullptr, MDVT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (flags != 0) return fail(nullptr, MDVT_ERR_INVALID_ARG, "flags must be 0");
    if (width < 1 || height < 1 || width > 65535 || height > 32767)
        return fail(nullptr, MDVT_ERR_INVALID_ARG, "frame size %dx%d out of range (1..65535 x 1..32767)", width, height);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, MDVT_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, MDVT_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, count - 1);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, MDVT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MDVT_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    mdvt_ctx* c = new (std::nothrow) mdvt_ctx();
    if (!c) return fail(nullptr, MDVT_ERR_OOM, "out of host memory");
    c->device = device; c->W = width; c->H = height;
    c->pool_tag = device;
    if (const char* t = tuning_env(TUNE_POOL_TAG)) c->pool_tag = atoi(t);
    { const char* e = getenv("MDVT_MESH_CONV"); c->opt_mesh_conv = e && e[0] == '1'; }     // the library's one switch, read here and nowhere else
    c->cfg.mode = MDVT_MODE_POINTS; c->cfg.ipd_m = 0.063; c->cfg.max_depth = 100.0;   // argparse defaults (sr:284, 288)
    *out = c;
    return MDVT_OK;
}

// ---- The banks' side stream and events: process-wide, never destroyed ------------------------------------------------------
// r05: three soak processes in ~3 000 multi-frame sweep jobs (300 k contexts that had used banks) died of a signal -- never one of
// 3 500 single-frame jobs (1.4 M contexts) -- and the native backtrace of the third (tools/probe/segv_trace.c) is the HSA runtime's
// own callback thread faulting inside libamdhip64, not a frame of this library.  What only the bank path has is a stream and four
// events created by a context and destroyed with it; whatever the runtime's handler still holds of them after the
// hipDeviceSynchronize of mdvt_destroy, the library no longer destroys them: they wait here for the next context of the same GPU
// that uses banks (the treatment the pinned parameter blocks and the workspace blocks got for their own reasons).
struct BankRes { hipStream_t side; hipEvent_t ev[4]; int device; };
static std::mutex g_bank_mutex;
static std::vector<BankRes>& bank_pool() { static std::vector<BankRes>* p = new std::vector<BankRes>(); return *p; }      // (leaked on purpose)

static hipError_t bank_res_take(mdvt_ctx* c)
{
    {
        std::lock_guard<std::mutex> lock(g_bank_mutex);
        auto& pool = bank_pool();
        for (size_t k = pool.size(); k-- > 0;)
            if (pool[k].device == c->device) {
                c->side = pool[k].side; c->ev_start = pool[k].ev[0]; c->ev_join = pool[k].ev[1]; c->ev_vert[0] = pool[k].ev[2]; c->ev_vert[1] = pool[k].ev[3];
                pool.erase(pool.begin() + (long)k);
                return hipSuccess;
            }
    }
    // built in locals, handed to the context only when complete: a half-made set must not leave c->side set (every later banked
    // render would skip this function and record a null event); what was made of it is abandoned, like everything of this kind
    BankRes r{};
    hipError_t e = hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking);
    for (hipEvent_t& ev : r.ev)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    c->side = r.side; c->ev_start = r.ev[0]; c->ev_join = r.ev[1]; c->ev_vert[0] = r.ev[2]; c->ev_vert[1] = r.ev[3];
    return hipSuccess;
}
// (the caller has synchronised the device: nothing submitted still uses them)
static void bank_res_give(mdvt_ctx* c)
{
    if (!c->side) return;
    if (c->ev_start && c->ev_join && c->ev_vert[0] && c->ev_vert[1]) {
        std::lock_guard<std::mutex> lock(g_bank_mutex);
        bank_pool().push_back({c->side, {c->ev_start, c->ev_join, c->ev_vert[0], c->ev_vert[1]}, c->device});
    }       // (a half-created set -- an error in bank_res_take -- is abandoned, not destroyed)
    c->side = nullptr; c->ev_start = c->ev_join = c->ev_vert[0] = c->ev_vert[1] = nullptr;
}

static void free_telea(mdvt_ctx* c)
{
    mdvt::TeleaWorkspace& w = c->telea;
    void* ptrs[] = {w.stamp, w.T, w.img, w.need, w.nlist, w.counts, w.remaining, w.last_round};   // offs / ncounts live inside counts
    for (void* p : ptrs) if (p) ws_free(c, p);
    w = mdvt::TeleaWorkspace{};
    c->telea_images = 0; c->telea_rounds = 0;
}

int mdvt_destroy(mdvt_ctx* c)
{
    if (!c) return MDVT_OK;
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    bank_res_give(c);
    if (c->hugeq2) ws_free(c, c->hugeq2);
    for (auto& sl : c->slots) {
        pool_give(sl.host, sl.dev, sl.capacity * sizeof(FrameDev), c->pool_tag);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (int e = 0; e < 2; ++e) { if (c->keys[e]) ws_free(c, c->keys[e]); if (c->ekeys[e]) ws_free(c, c->ekeys[e]); if (c->cbuf[e]) ws_free(c, c->cbuf[e]); }
    if (c->bigq) ws_free(c, c->bigq);
    if (c->tri_invalid) ws_free(c, c->tri_invalid);
    if (c->unused) ws_free(c, c->unused);
    if (c->elist) ws_free(c, c->elist);
    if (c->row_counts) ws_free(c, c->row_counts);
    if (c->wave_counts) ws_free(c, c->wave_counts);
    if (c->divcheck) ws_free(c, c->divcheck);
    if (c->rowcell) ws_free(c, c->rowcell);
    pool_give(c->telea_levels_host, nullptr, 64, -1);
    free_telea(c);
    if (c->ni_ws) ws_free(c, c->ni_ws);
    delete c;
    return MDVT_OK;
}

int mdvt_set_config(mdvt_ctx* c, const mdvt_config* cfg)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!cfg) return fail(c, MDVT_ERR_INVALID_ARG, "cfg is NULL");
    if (cfg->mode != MDVT_MODE_POINTS && cfg->mode != MDVT_MODE_MESH) return fail(c, MDVT_ERR_INVALID_ARG, "unknown mode %d", cfg->mode);
    if (!(cfg->max_depth > 0.0)) return fail(c, MDVT_ERR_INVALID_ARG, "max_depth must be > 0");
    if (!(cfg->ipd_m >= 0.0)) return fail(c, MDVT_ERR_INVALID_ARG, "ipd_m must be >= 0");
    if (cfg->edge_points < 0 || cfg->edge_points > 2) return fail(c, MDVT_ERR_INVALID_ARG, "edge_points must be 0, 1 or 2");
    if (cfg->edge_points && !cfg->remove_edges) return fail(c, MDVT_ERR_INVALID_ARG, "edge_points needs remove_edges (sr:589)");
    if (cfg->cull < 0 || cfg->cull > 2) return fail(c, MDVT_ERR_INVALID_ARG, "cull must be 0 (none), 1 (back) or 2 (front)");
    if (cfg->subpixel_bits != 0 && cfg->subpixel_bits != 4 && cfg->subpixel_bits != 8)
        return fail(c, MDVT_ERR_INVALID_ARG, "subpixel_bits must be 0 (default: 8), 4 or 8 -- the grids this build's rasterisers are compiled for");
    if (cfg->reserved2 != 0) return fail(c, MDVT_ERR_INVALID_ARG, "reserved2 must be 0");
    // (advisor r04: the field took over a reserved one -- a caller built against 0.11 that left it uninitialised must not get an
    //  arbitrary budget silently: anything above 1 TiB is refused, small values are honoured down to one slot)
    if (cfg->workspace_mib > (1u << 20)) return fail(c, MDVT_ERR_INVALID_ARG, "workspace_mib %u out of range (0 = default 4096, at most 1048576)", cfg->workspace_mib);
    c->cfg = *cfg;
    c->cfg_set = true;
    return MDVT_OK;
}

int mdvt_selftest(mdvt_ctx* c, int which, uint64_t seed, uint64_t* h_mismatches)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!h_mismatches || which < 0 || which > 2) return fail(c, MDVT_ERR_INVALID_ARG, "mdvt_selftest: which must be 0..2, h_mismatches not NULL");
    DeviceGuard g(c->device);
    unsigned long long* d = nullptr;
    MDVT_HIP(c, hipMalloc((void**)&d, sizeof(unsigned long long)));
    hipError_t e = launch_selftest(which, (unsigned long long)seed, d, nullptr);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, MDVT_ERR_HIP, "mdvt_selftest: %s", hipGetErrorString(e));
    *h_mismatches = (uint64_t)h;
    return MDVT_OK;
}

int mdvt_render_stereo_batch(mdvt_ctx* c, int n_frames, const mdvt_frame_params* params, const mdvt_io* io, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (n_frames <= 0 || !params || !io) return fail(c, MDVT_ERR_INVALID_ARG, "n_frames/params/io invalid");
    if (!io->depth_rgb || !io->color_rgb || !io->left_rgb || !io->right_rgb)
        return fail(c, MDVT_ERR_INVALID_ARG, "depth_rgb, color_rgb and left/right rgb buffers are required");
    // The byte masks may be left out (both NULL) by a caller that takes the packed mask instead -- where the compaction is fused
    // into the render kernel (pure-shift point frames: checked per run below); everywhere else they are required.
    const bool no_byte_mask = !io->left_mask && !io->right_mask && io->left_maskbits && io->right_maskbits;
    if (!no_byte_mask && (!io->left_mask || !io->right_mask))
        return fail(c, MDVT_ERR_INVALID_ARG, "left/right mask buffers are required (both may be NULL only when maskbits are given)");
    const int W = c->W, H = c->H;
    if (W < 2 || H < 2) return fail(c, MDVT_ERR_INVALID_ARG, "rendering needs at least a 2x2 frame");
    if (io->depth_pitch < (size_t)3 * W || io->color_pitch < (size_t)3 * W || io->rgb_pitch < (size_t)3 * W ||
        (!no_byte_mask && io->mask_pitch < (size_t)W))
        return fail(c, MDVT_ERR_INVALID_ARG, "a pitch is smaller than one row");   // sr:507 shape assert
    const bool zout = io->left_depth || io->right_depth;
    if (zout && io->zout_pitch < (size_t)4 * W) return fail(c, MDVT_ERR_INVALID_ARG, "zout_pitch smaller than one row");
    if (io->left_seed || io->right_seed) {
        if (!io->left_seed || !io->right_seed) return fail(c, MDVT_ERR_INVALID_ARG, "seed images need both eyes");
        if (!c->cfg.remove_edges) return fail(c, MDVT_ERR_INVALID_ARG, "seed images need remove_edges (the infill-mask mode of sr:568-570)");
        if (io->seed_pitch < (size_t)3 * W) return fail(c, MDVT_ERR_INVALID_ARG, "seed_pitch smaller than one row");
    }
    const bool want_bits = io->left_maskbits || io->right_maskbits;
    if (want_bits) {
        if (!io->left_maskbits || !io->right_maskbits) return fail(c, MDVT_ERR_INVALID_ARG, "maskbits need both eyes");
        if (io->maskbits_pitch < (size_t)4 * (((size_t)W + 31) / 32) || io->maskbits_pitch % 4 != 0 || io->maskbits_stride % 4 != 0 ||
            ((uintptr_t)io->left_maskbits % 4) || ((uintptr_t)io->right_maskbits % 4))
            return fail(c, MDVT_ERR_INVALID_ARG, "maskbits rows must be dword aligned and at least 4*ceil(W/32) bytes");
    }
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;

    std::vector<FrameDev> fd((size_t)n_frames);
    int general = 0;
    for (int k = 0; k < n_frames; ++k) {
        const int rc = fill_frame_dev(c, params[k], fd[(size_t)k]);
        if (rc != MDVT_OK) return rc;
        general |= fd[(size_t)k].general;
        fd[(size_t)k].div_slot = -1;
    }
    // Pure-shift point frames: the disparity's division proven short per parameter set (FrameDev.div_slot).  A new set costs one
    // launch of 65536 threads on this stream, once per context; clips have one set, or one per distinct field of view.
    if (c->cfg.mode == MDVT_MODE_POINTS) {
        for (int k = 0; k < n_frames; ++k) {
            FrameDev& f = fd[(size_t)k];
            if (f.general) continue;
            std::array<uint32_t, 3> key;
            memcpy(&key[0], &f.mult, 4); memcpy(&key[1], &f.scale, 4); memcpy(&key[2], &f.dl, 4);
            int slot = -1;
            for (size_t q = c->div_keys.size(); q-- > 0;) if (c->div_keys[q] == key) { slot = (int)q; break; }
            if (slot < 0 && c->div_keys.size() < (size_t)mdvt::kDivSlots) {
                if (!c->divcheck) {
                    MDVT_HIP(c, ws_malloc(c, (void**)&c->divcheck, mdvt::kDivSlots * sizeof(uint32_t), s));
                    MDVT_HIP(c, hipMemsetAsync(c->divcheck, 0, mdvt::kDivSlots * sizeof(uint32_t), s));
                }
                slot = (int)c->div_keys.size();
                MDVT_HIP(c, MDVT_GRID_CALL(c, launch_divcheck, f.mult, f.scale, f.dl, c->divcheck + slot, s));
                c->div_keys.push_back(key);
            }
            f.div_slot = slot;
        }
    }
    // The arithmetic of a frame (pure shift or general) is its own property, never its batch
    // neighbours': consecutive frames of one kind form a run, every run gets its own launches.
    struct Run { int f0, f1, general, conv, craster; };  // general = "takes the global-key kernels"; conv = k_mesh_conv (mesh, convergence only);
                                                         // craster = general, but every frame convergence-only: k_mesh_raster_conv
    std::vector<Run> runs;

    const FrameDev* dfp = nullptr;
    ParamSlot* slot = nullptr;
    int rc = stage_params(c, fd, s, &dfp, &slot);
    if (rc != MDVT_OK) return rc;

    RenderPlan plan{};
    plan.mode = c->cfg.mode;
    plan.remove_edges = c->cfg.remove_edges;
    plan.edge_points = c->cfg.remove_edges && c->cfg.edge_points;
    plan.general = general;
    plan.allow_conv = c->opt_mesh_conv ? 1 : 0;
    if (tuning_build()) { const char* e = tuning_env(TUNE_MESH_CONV); plan.allow_conv = (e && e[0] == '1') ? 1 : 0; }   // (tests toggle it per call)
    plan.vec4 = (W % 4 == 0) && aligned(io->depth_rgb, 4) && aligned(io->color_rgb, 4) && aligned(io->left_rgb, 4) &&
                aligned(io->right_rgb, 4) && aligned(io->left_mask, 4) && aligned(io->right_mask, 4) &&
                io->depth_pitch % 4 == 0 && io->color_pitch % 4 == 0 && io->rgb_pitch % 4 == 0 && io->mask_pitch % 4 == 0 &&
                io->depth_stride % 4 == 0 && io->color_stride % 4 == 0 && io->rgb_stride % 4 == 0 && io->mask_stride % 4 == 0 &&
                (!io->left_seed || (aligned(io->left_seed, 4) && aligned(io->right_seed, 4) && io->seed_pitch % 4 == 0 && io->seed_stride % 4 == 0)) &&
                (!zout || ((!io->left_depth || aligned(io->left_depth, 16)) && (!io->right_depth || aligned(io->right_depth, 16)) &&
                           io->zout_pitch % 16 == 0 && io->zout_stride % 16 == 0));

    // A pure-shift frame wider than the LDS row kernels can hold (10 240 px for points, ~4 300 for the mesh with edge
    // points) is rendered by the global-key kernels instead -- with its own pure-shift arithmetic (FrameDev.general
    // stays 0), so the pixels do not depend on which kernels ran.  MDVT_FORCE_GLOBAL=1 sends every frame that way (tests).
    const bool wide = !MDVT_GRID_CALL(c, render_fits_lds, plan, W) || tuning_env(TUNE_FORCE_GLOBAL) != nullptr;
    if (wide) general = 1;
    bool conv_kernel = false;
    if (plan.mode == MDVT_MODE_MESH && !wide) {
        RenderArgs probe{};
        probe.W = W; probe.H = H;
        conv_kernel = MDVT_GRID_CALL(c, mesh_conv_supported, plan, probe);
    }
    bool any_global = false, any_conv = false;
    for (int k = 0; k < n_frames; ++k) {
        const int cv = (conv_kernel && fd[(size_t)k].conv_band) ? 1 : 0;
        const int g = (!cv && (wide || fd[(size_t)k].general || fd[(size_t)k].erow_wild)) ? 1 : 0;
        const int cr = (g && !wide && plan.mode == MDVT_MODE_MESH && fd[(size_t)k].conv_band) ? 1 : 0;
        any_global |= g != 0; any_conv |= cv != 0;
        if (runs.empty() || runs.back().general != g || runs.back().conv != cv || runs.back().craster != cr) runs.push_back({k, k + 1, g, cv, cr});
        else runs.back().f1 = k + 1;
    }
    general = (any_global || any_conv) ? 1 : 0;          // some run uses the global workspace
    const bool need_keys = any_global;
    const bool need_ekeys = general && plan.edge_points;
    const bool need_mesh_ws = any_global && plan.mode == MDVT_MODE_MESH;
    // frames per launch set.  Point splat, general: two frames keep the 64-bit key buffers (33 MB per 1080p frame)
    // inside the 256 MiB Infinity Cache between splat and resolve (measured +12 %); the mesh needs the slack of
    // eight (rows full of slivers leave a long tail), and the edge filter alone streams, so 8 as well.
    int tuned_chunk = 0;
    if (const char* e = tuning_env(TUNE_WS_CHUNK)) { const int v = atoi(e); if (v > 0) tuned_chunk = v; }   // tuning hook
    auto chunk_of = [&](const Run& r) {
        const int n = r.f1 - r.f0;
        if (!(r.general || plan.remove_edges || (r.conv && plan.edge_points))) return n;                  // no workspace: the whole run in one launch
        // (points, general path: four slots -- one launch set of four frames, or banks of two (below); two slots until r04:
        //  1080p convergence 26.7 k -> 28.2 k frames/s, 4K pose + contention 5.4 k -> 6.1 k)
        int ws_chunk = (r.general && plan.mode == MDVT_MODE_POINTS) ? 4 : kWorkspaceChunk;
        if (r.general && plan.mode == MDVT_MODE_MESH) {
            ws_chunk = 2 * kWorkspaceChunk;      // 16: measured -4 % (convergence) / -11 % (pose) vs 8
            // ~64 B/px per slot (z keys, tie side words, triangle queue; until r04 also 32 B/px of vertex records): 2.1 GB at 1080p,
            // 8.5 GB at 4K; the queue's entry indices are 32-bit, so very large frames get fewer slots (4 entries per pixel and slot)
            const size_t fit = (size_t)0xFFFFFFF0u / (4 * (size_t)W * (size_t)H);
            if ((size_t)ws_chunk > fit) ws_chunk = fit < 1 ? 1 : (int)fit;
            // ... and the slots have to fit the context's workspace budget (mdvt_config.workspace_mib, default 4 GiB: 16 slots at
            // 1080p, 8 at 3840 x 2160 -- where 16 would be 8.5 GB): per slot and pixel 16 B of z keys, 16 B of tie side words,
            // 32 B of triangle queue, with edge points 28 B of edge keys, their list and the vertex list, 3 B of filter flags
            const size_t per_slot = (size_t)W * (size_t)H * (16 + 16 + 32 + (plan.edge_points ? 28 : 0) + (plan.remove_edges ? 3 : 0));
            const size_t budget = (size_t)(c->cfg.workspace_mib ? c->cfg.workspace_mib : 4096u) << 20;
            const size_t afford = budget / per_slot;
            if ((size_t)ws_chunk > afford) ws_chunk = afford < 1 ? 1 : (int)afford;
        }
        // pure-shift mesh rows with edge removal: a launch is (frames x 135 bands) workgroups for 512 slots -- 8 frames
        // leave the chip 30 % idle in the last wave of workgroups (476 -> see DESIGN.md); the workspace is 11 B/px per frame
        // (points with edge removal likewise since r04: every launch set ends with k_edge_rows_exact, a handful of workgroups the
        //  stream waits for -- once per 32 frames instead of once per 8)
        if (!r.general) ws_chunk = 4 * kWorkspaceChunk;
        if (tuned_chunk) ws_chunk = tuned_chunk;
        if (r.general && ws_chunk > 32) ws_chunk = 32;        // one parity bit per z-key slot (uint32_t key_parity)
        return n < ws_chunk ? n : ws_chunk;
    };
    int ws_frames = 0, count_frames = 0;
    for (const Run& r : runs) {
        const int ch = chunk_of(r);
        if ((r.general || r.conv || plan.remove_edges) && ch > ws_frames) ws_frames = ch;
        if (ch > count_frames) count_frames = ch;
    }
    if (ws_frames && (rc = ensure_workspace(c, ws_frames, need_keys, need_ekeys, plan.remove_edges, need_mesh_ws, s)) != MDVT_OK) return rc;

    RenderArgs a{};
    a.depth = io->depth_rgb; a.depth_pitch = io->depth_pitch; a.depth_stride = io->depth_stride;
    a.color = io->color_rgb; a.color_pitch = io->color_pitch; a.color_stride = io->color_stride;
    a.rgb[0] = io->left_rgb; a.rgb[1] = io->right_rgb; a.rgb_pitch = io->rgb_pitch; a.rgb_stride = io->rgb_stride;
    a.mask[0] = io->left_mask; a.mask[1] = io->right_mask; a.mask_pitch = io->mask_pitch; a.mask_stride = io->mask_stride;
    a.zout[0] = io->left_depth; a.zout[1] = io->right_depth; a.zout_pitch = io->zout_pitch; a.zout_stride = io->zout_stride;
    a.maskbits[0] = io->left_maskbits; a.maskbits[1] = io->right_maskbits;
    a.maskbits_pitch = io->maskbits_pitch; a.maskbits_stride = io->maskbits_stride;
    a.hole_counts = io->hole_counts;
    a.seed[0] = io->left_seed; a.seed[1] = io->right_seed; a.seed_pitch = io->seed_pitch; a.seed_stride = io->seed_stride;
    if (io->hole_counts) {
        if (c->row_counts_frames < count_frames) {
            if (c->row_counts) { MDVT_HIP(c, hipDeviceSynchronize()); ws_free(c, c->row_counts); ws_free(c, c->wave_counts); }     // (earlier submissions may still count into it)
            c->row_counts = nullptr; c->wave_counts = nullptr; c->row_counts_frames = 0;
            MDVT_HIP(c, ws_malloc(c, (void**)&c->row_counts, (size_t)count_frames * 2 * H * sizeof(uint32_t), s));
            MDVT_HIP(c, ws_malloc(c, (void**)&c->wave_counts, (size_t)count_frames * H * 16 * sizeof(uint32_t), s));
            c->row_counts_frames = count_frames;
        }
        a.row_counts = c->row_counts;
        a.wave_counts = c->wave_counts;
    }
    a.fp = dfp;
    a.divcheck = c->divcheck;
    a.edge_paint = c->cfg.edge_points != 2;
    a.cull = c->cfg.cull;
    if (c->cfg.mode == MDVT_MODE_MESH) { if ((rc = ensure_rowcell(c, s)) != MDVT_OK) return rc; a.rowcell = c->rowcell; }
    a.W = W; a.H = H;
    a.key_rgb = (uint32_t)c->cfg.key_rgb[0] | ((uint32_t)c->cfg.key_rgb[1] << 8) | ((uint32_t)c->cfg.key_rgb[2] << 16);
    a.keys[0] = c->keys[0]; a.keys[1] = c->keys[1];
    a.ekeys[0] = c->ekeys[0]; a.ekeys[1] = c->ekeys[1];
    a.elist = c->elist; a.elist_count = c->elist ? c->elist + (size_t)c->ws_frames * 2 * (size_t)W * H : nullptr;
    a.vlist = c->elist ? a.elist_count + (size_t)c->ws_frames * H : nullptr;
    a.vlist_count = c->elist ? a.vlist + (size_t)c->ws_frames * (size_t)W * H : nullptr;
    a.cbuf[0] = c->cbuf[0]; a.cbuf[1] = c->cbuf[1];
    a.tri_invalid = c->tri_invalid; a.unused = c->unused;
    if (c->bigq) {
        a.bigq = c->bigq; a.bigq_cap = c->bigq_cap; a.bigq_count = c->bigq + (size_t)c->bigq_cap * mdvt::kBigRecDwords;
        a.hugeq = a.bigq_count + 2 * (size_t)c->ws_frames * H + 8;      // (8-byte aligned: entries are uint2); two lists (banks, below)
        a.tie_flag = a.hugeq + (size_t)c->huge_lists * (2 * (size_t)mdvt::kHugeCap + 2);
        a.tie_tiles = a.tie_flag + c->ws_frames;
        a.tie_words = (int32_t)mdvt::tie_words_of(W, H);
        a.tie_tiles_x = (W + mdvt::kTieTile - 1) / mdvt::kTieTile;
    }
    a.ws_stride_px = (size_t)W * H;
    a.ws_stride_tri = 2 * (size_t)(W - 1) * (H - 1);

    if (general) {
        if (c->keys_dirty) {        // re-establish the EMPTY invariant the resolve pass normally maintains
            const size_t bytes = (size_t)c->ws_frames * a.ws_stride_px * sizeof(unsigned long long);
            for (int e = 0; e < 2; ++e) {
                if (c->keys[e]) MDVT_HIP(c, hipMemsetAsync(c->keys[e], 0xFF, bytes, s));
                if (c->ekeys[e]) MDVT_HIP(c, hipMemsetAsync(c->ekeys[e], 0xFF, bytes, s));
            }
            if (c->elist) MDVT_HIP(c, hipMemsetAsync(a.elist_count, 0, (size_t)c->ws_frames * H * sizeof(uint32_t), s));
            c->key_parity = 0;
        }
        c->keys_dirty = true;
    }
    for (const Run& r : runs) {
      plan.general = r.general;
      plan.conv = r.conv;
      plan.conv_raster = r.craster;
      int chunk = chunk_of(r);
      // Posed / converged mesh frames in more than one launch set: the sets take turns on two halves ("banks") of the workspace slots
      // and on two streams, a set starting when the vertex pass of the set before it is through -- the path's stages wait for
      // different things (the vertex pass for its stores, the rasteriser for its atomics), and the next set's vertex pass and edge
      // filter fill the rasteriser's waits: 32 frames of 1080p product default +3 %, mesh + convergence +4 %, mesh under a pose +7 %,
      // 8 frames of 4K pose + contention (C4) +10 %; a run that fits ONE launch set stays as it is (16 frames: two sets of 8 lose 1.5 %).
      // Points on the general path likewise (splat, then resolve: the next set's splat beside this set's resolve): C4 points +13 %.
      const bool bankable = r.general && !r.conv && !want_bits && !io->hole_counts && tuning_env(TUNE_WS_CHUNK) == nullptr;
      bool banks = bankable && chunk >= 2 && r.f1 - r.f0 > chunk;
      int bank_slots = chunk / 2;
      // r05: a posed mesh run that FITS one launch set is split into two sets on the two banks all the same when each half is large
      // enough to fill the chip by itself (3 frames of 4K = 24.9 M pixels: 6 to 8 frames of C4's shape) -- since the vertex records went (64 B/px per slot, was 96) the
      // 8 frames of C4 are one set of 8 slots, and its cell walk (VALU) and resolve (HBM) ran one after the other again
      if (bankable && !banks && plan.mode == MDVT_MODE_MESH && r.f1 - r.f0 <= chunk && r.f1 - r.f0 >= 4 &&
          (size_t)((r.f1 - r.f0) / 2) * (size_t)W * (size_t)H >= (size_t)3 * 3840 * 2160) {
          banks = true;
          bank_slots = (r.f1 - r.f0) / 2;
      }
      hipStream_t const s_call = s;
      // (every way out of the bank loop joins the side stream back into the caller's: an error return must not leave the side
      //  stream working on its half of the workspace -- and on the caller's output buffers -- behind the caller's back; advisor, r04)
      struct BankJoin {
          mdvt_ctx* c; hipStream_t s_call; bool armed;
          ~BankJoin() {
              if (!armed) return;
              if (hipEventRecord(c->ev_join, c->side) != hipSuccess || hipStreamWaitEvent(s_call, c->ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(c->side);
          }
      } bank_join{c, s_call, false};
      if (banks) {
          chunk = bank_slots;
          if (!c->side) MDVT_HIP(c, bank_res_take(c));        // (process-wide: see bank_res_take)
          MDVT_HIP(c, hipEventRecord(c->ev_start, s_call));            // (the inputs, the parameter block, the runs before this one)
          MDVT_HIP(c, hipStreamWaitEvent(c->side, c->ev_start, 0));
          bank_join.armed = true;
      }
      const RenderArgs a_all = a;
      int set = 0;
      for (int f0 = r.f0; f0 < r.f1; f0 += chunk, ++set) {
        plan.n = (r.f1 - f0 < chunk) ? r.f1 - f0 : chunk;
        const int bank = banks ? (set & 1) : 0, slot0 = bank * bank_slots;
        hipStream_t s = bank ? c->side : s_call;
        if (banks) {
            a = a_all;
            const size_t px0 = (size_t)slot0 * a.ws_stride_px;
            for (int e = 0; e < 2; ++e) {
                a.keys[e] += px0;
                if (a.cbuf[e]) a.cbuf[e] += px0;
                if (a.ekeys[e]) a.ekeys[e] += px0;
            }
            if (a.elist) { a.elist += (size_t)slot0 * 2 * (size_t)W * H; a.elist_count += (size_t)slot0 * H; a.vlist += px0; a.vlist_count += slot0; }
            if (a.tri_invalid) a.tri_invalid += (size_t)slot0 * a.ws_stride_tri;
            if (a.unused) a.unused += px0;
            if (a.bigq) {
                a.bigq += (size_t)slot0 * H * (size_t)(4 * W) * mdvt::kBigRecDwords;
                a.bigq_count += (size_t)bank * ((2 * (size_t)bank_slots * H + 2 + 3) & ~(size_t)3);      // (counters and prefix sums of a set; 16-byte aligned)
                // The second bank's own huge list: a separate allocation, made when banks are first used (r04: with both lists in the
                // queue's block a 100 x 31 frame's block passed 2 MB and left the runtime's fragment cache -- see the workspace pool above).
                if (bank && c->huge_lists == 2) a.hugeq += 2 * (size_t)mdvt::kHugeCap + 2;
                else if (bank) {
                    if (!c->hugeq2) MDVT_HIP(c, ws_malloc(c, (void**)&c->hugeq2, (2 * (size_t)mdvt::kHugeCap + 2) * sizeof(uint32_t), s));
                    a.hugeq = c->hugeq2;
                }
                a.tie_flag += slot0;
                a.tie_tiles += (size_t)slot0 * 2 * a.tie_words;
            }
            if (set > 0) MDVT_HIP(c, hipStreamWaitEvent(s, c->ev_vert[bank ^ 1], 0));      // (the set before this one has projected its vertices / splatted its points)
            plan.after_vertices = c->ev_vert[bank];
        }
        a.frame0 = f0;
        if (plan.remove_edges) {
            MDVT_HIP(c, launch_zero_bytes(a.unused, (size_t)plan.n * a.ws_stride_px, s));
            MDVT_HIP(c, launch_edge_filter(a.depth, a.depth_pitch, a.depth_stride, dfp, f0, plan.n, W, H,
                                           plan.mode == MDVT_MODE_MESH, a.tri_invalid, a.ws_stride_tri,
                                           a.unused, a.ws_stride_px, s));
        }
        if (no_byte_mask && !MDVT_GRID_CALL(c, points_fused_bits_applies, plan, a))
            return fail(c, MDVT_ERR_INVALID_ARG, "the byte masks may be NULL only where the mask compaction is fused into the render "
                        "(points mode, pure stereo shift, no edge removal, W %% 4 == 0, W <= 4096)");
        a.key_parity = c->key_parity >> slot0;
        plan.edge_rows_max = 0;
        if (!r.general && !r.conv && plan.edge_points)
            for (int k = f0; k < f0 + plan.n; ++k)
                if (fd[(size_t)k].erow_lo < fd[(size_t)k].erow_hi)
                    plan.edge_rows_max = std::max(plan.edge_rows_max, fd[(size_t)k].erow_hi - fd[(size_t)k].erow_lo + 1);
        hipError_t e = MDVT_GRID_CALL(c, launch_render, plan, a, s);
        plan.after_vertices = nullptr;
        if (r.general && e == hipSuccess) c->key_parity ^= (plan.n >= 32 ? 0xFFFFFFFFu : ((1u << plan.n) - 1u)) << slot0;   // these slots' next use has the other parity
        if (e == hipErrorNotSupported) return fail(c, MDVT_ERR_UNSUPPORTED, "render mode %d is not built yet", plan.mode);
        if (e != hipSuccess) return fail(c, MDVT_ERR_HIP, "render launch failed: %s", hipGetErrorString(e));
        if ((want_bits || io->hole_counts) && !plan.fused_bits) MDVT_HIP(c, launch_pack_mask(a, plan.n, s));
        if (io->hole_counts && !plan.fused_bits) MDVT_HIP(c, launch_reduce_counts(a, plan.n, s));
      }
      if (banks) {
          a = a_all;
          MDVT_HIP(c, hipEventRecord(c->ev_join, c->side));
          MDVT_HIP(c, hipStreamWaitEvent(s_call, c->ev_join, 0));
          bank_join.armed = false;
      }
    }
    if (general) c->keys_dirty = false;
    MDVT_HIP(c, hipEventRecord(slot->done, s));
    return MDVT_OK;
}

int mdvt_render_stereo(mdvt_ctx* c, const mdvt_frame_params* params, const mdvt_io* io, void* stream)
{
    return mdvt_render_stereo_batch(c, 1, params, io, stream);
}

int mdvt_decode_depth(mdvt_ctx* c, const uint8_t* d_rgb, size_t rgb_pitch, float* d_depth, size_t depth_pitch,
                      double max_depth, double depth_scale, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_rgb || !d_depth) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (rgb_pitch < (size_t)3 * c->W || depth_pitch < (size_t)4 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (!(max_depth > 0.0)) return fail(c, MDVT_ERR_INVALID_ARG, "max_depth must be > 0");
    DeviceGuard g(c->device);
    MDVT_HIP(c, launch_decode_depth(d_rgb, rgb_pitch, d_depth, depth_pitch, c->W, c->H,
                                    (float)(max_depth / 4228250625.0), (float)depth_scale, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_encode_depth(mdvt_ctx* c, const float* d_depth, size_t depth_pitch, uint8_t* d_rgb, size_t rgb_pitch,
                      double max_depth, int bgr, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_rgb || !d_depth) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (rgb_pitch < (size_t)3 * c->W || depth_pitch < (size_t)4 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (!(max_depth > 0.0)) return fail(c, MDVT_ERR_INVALID_ARG, "max_depth must be > 0");
    DeviceGuard g(c->device);
    MDVT_HIP(c, launch_encode_depth(d_depth, depth_pitch, d_rgb, rgb_pitch, c->W, c->H, max_depth, bgr, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_release_cached_memory(int device)
{
    std::vector<DevBlock> out;
    {
        std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
        auto& pool = dev_pool();
        for (size_t k = 0; k < pool.size();) {
            if (device >= 0 && pool[k].tag != device) { ++k; continue; }
            out.push_back(pool[k]);
            dev_pool_idle()[pool[k].tag] -= pool[k].bytes;
            pool.erase(pool.begin() + (long)k);
        }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    for (const DevBlock& b : out) {
        // (a block tagged for a GPU this process does not have -- the tuning build's MDVT_POOL_TAG -- lives on the current one)
        DeviceGuard g(b.tag >= 0 && b.tag < count ? b.tag : 0);
        (void)hipDeviceSynchronize();
        (void)hipFree(b.p);
    }
    return MDVT_OK;
}

int mdvt_set_cached_memory_limit(uint64_t bytes_per_gpu)
{
    std::vector<DevBlock> out;
    {
        std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
        g_dev_pool_idle_cap = (size_t)bytes_per_gpu;
        auto& pool = dev_pool();
        for (size_t k = 0; k < pool.size();) {                                           // oldest first, per GPU
            size_t& idle = dev_pool_idle()[pool[k].tag];
            if (idle <= g_dev_pool_idle_cap) { ++k; continue; }
            out.push_back(pool[k]);
            idle -= pool[k].bytes;
            pool.erase(pool.begin() + (long)k);
        }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    for (const DevBlock& b : out) {
        DeviceGuard g(b.tag >= 0 && b.tag < count ? b.tag : 0);
        (void)hipDeviceSynchronize();
        (void)hipFree(b.p);
    }
    return MDVT_OK;
}

int mdvt_cached_memory(int device, uint64_t* idle_bytes, uint64_t* idle_blocks)
{
    uint64_t bytes = 0, blocks = 0;
    {
        std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
        for (const DevBlock& b : dev_pool()) if (device < 0 || b.tag == device) { bytes += b.bytes; ++blocks; }
    }
    if (idle_bytes) *idle_bytes = bytes;
    if (idle_blocks) *idle_blocks = blocks;
    return MDVT_OK;
}

int mdvt_debug_read(mdvt_ctx* c, int what, void* h_dst, uint64_t capacity, uint64_t info[8])
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!tuning_build()) return fail(c, MDVT_ERR_UNSUPPORTED, "mdvt_debug_read: tuning build only");
    if (what < 0 || what > 2 || !info) return fail(c, MDVT_ERR_INVALID_ARG, "mdvt_debug_read: what must be 0, 1 or 2, info not NULL");
    if (what == 2) {
        // the two process-wide pools as this context's GPU sees them (its pool tag: the device, or MDVT_POOL_TAG): idle parameter
        // blocks that carry device memory of this / of another GPU, idle workspace blocks of this / of another GPU
        for (int k = 0; k < 8; ++k) info[k] = 0;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            for (const PoolBlock& b : param_pool()) if (b.device >= 0) ++info[b.device == c->pool_tag ? 0 : 1];
        }
        {
            std::lock_guard<std::mutex> lock(g_dev_pool_mutex);
            for (const DevBlock& b : dev_pool()) ++info[b.tag == c->pool_tag ? 2 : 3];
        }
        info[4] = (uint64_t)c->pool_tag;
        return MDVT_OK;
    }
    DeviceGuard g(c->device);
    MDVT_HIP(c, hipDeviceSynchronize());
    if (what == 1) {
        // the coherence test of mdvt_selftest.hip on the queue block itself (it OVERWRITES the block: the next render rewrites what it
        // reads): h_dst receives 80 dwords; info[0] = the tag used
        if (!c->bigq || !h_dst || capacity < 80 * sizeof(uint32_t)) return fail(c, MDVT_ERR_INVALID_ARG, "mdvt_debug_read: no queue block / 320 bytes needed");
        static uint32_t tag = 0x1234567u;
        tag = tag * 1664525u + 1013904223u;
        uint32_t *d_xcc = nullptr, *d_out = nullptr;
        MDVT_HIP(c, hipMalloc((void**)&d_xcc, (c->bigq_bytes / 256 + 1) * sizeof(uint32_t)));
        MDVT_HIP(c, hipMalloc((void**)&d_out, 80 * sizeof(uint32_t)));
        hipError_t e = launch_coherence_test(c->bigq, c->bigq_bytes / 4, tag, d_xcc, d_out, nullptr);
        if (e == hipSuccess) e = hipMemcpy(h_dst, d_out, 80 * sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void)hipFree(d_xcc); (void)hipFree(d_out);
        if (e != hipSuccess) return fail(c, MDVT_ERR_HIP, "mdvt_debug_read: %s", hipGetErrorString(e));
        info[0] = tag;
        return MDVT_OK;
    }
    const size_t cap_dw = (size_t)c->bigq_cap * mdvt::kBigRecDwords;
    info[0] = c->bigq ? c->bigq_bytes : 0;                                   // bytes of the queue block
    info[1] = cap_dw;                                                        // dword offset of the segment counters
    info[2] = (uint64_t)c->ws_frames * (uint64_t)c->H;                       // segments the block has room for
    info[3] = cap_dw + 2 * (uint64_t)c->ws_frames * (uint64_t)c->H + 8;      // dword offset of the (first) huge list
    info[4] = info[3] + (uint64_t)c->huge_lists * (2 * (uint64_t)mdvt::kHugeCap + 2);   // dword offset of the tie flags
    info[5] = (uint64_t)c->W; info[6] = (uint64_t)c->H; info[7] = (uint64_t)c->ws_frames;
    if (h_dst && c->bigq) {
        if (capacity < c->bigq_bytes) return fail(c, MDVT_ERR_INVALID_ARG, "mdvt_debug_read: %zu bytes needed", c->bigq_bytes);
        MDVT_HIP(c, hipMemcpy(h_dst, c->bigq, c->bigq_bytes, hipMemcpyDeviceToHost));
    }
    return MDVT_OK;
}

int mdvt_workspace_bytes(mdvt_ctx* c, uint64_t* bytes)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!bytes) return fail(c, MDVT_ERR_INVALID_ARG, "NULL argument");
    *bytes = (uint64_t)c->ws_bytes;
    return MDVT_OK;
}

int mdvt_edge_point_pixels(mdvt_ctx* c, const mdvt_frame_params* params, const uint8_t* d_depth_rgb, size_t depth_pitch,
                           int how, int32_t* d_px, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!c->cfg_set) return fail(c, MDVT_ERR_INVALID_ARG, "mdvt_set_config has not been called");
    if (!params || !d_depth_rgb || !d_px) return fail(c, MDVT_ERR_INVALID_ARG, "NULL argument");
    if (depth_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (how != 0 && how != 1) return fail(c, MDVT_ERR_INVALID_ARG, "how must be 0 (the chain) or 1 (as the row kernels take it)");
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    std::vector<FrameDev> fd(1);
    // (the row range is only worked out for configurations that splat edge points; this entry point always wants it)
    mdvt_config saved = c->cfg;
    c->cfg.remove_edges = 1; c->cfg.edge_points = 1;
    const int rc0 = fill_frame_dev(c, *params, fd[0]);
    c->cfg = saved;
    if (rc0 != MDVT_OK) return rc0;
    if (how == 1 && (fd[0].general || fd[0].erow_wild))
        return fail(c, MDVT_ERR_INVALID_ARG, "how = 1 needs a pure-shift frame whose rows the row kernels take");
    const FrameDev* dfp = nullptr;
    ParamSlot* slot = nullptr;
    const int rc = stage_params(c, fd, s, &dfp, &slot);
    if (rc != MDVT_OK) return rc;
    MDVT_HIP(c, launch_edge_point_pixels(d_depth_rgb, depth_pitch, dfp, c->W, c->H, c->cfg.mode == MDVT_MODE_MESH ? 1 : 0, how, d_px, s));
    MDVT_HIP(c, hipEventRecord(slot->done, s));
    return MDVT_OK;
}

int mdvt_edge_filter(mdvt_ctx* c, const uint8_t* d_depth_rgb, size_t depth_pitch, const double K[9], double depth_scale,
                     int of_by_one, uint8_t* d_tri_invalid, uint8_t* d_unused, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_depth_rgb || !K) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (depth_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (c->W < 2 || c->H < 2) return fail(c, MDVT_ERR_INVALID_ARG, "the edge filter needs at least a 2x2 frame");
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    std::vector<FrameDev> fd(1);
    FrameDev& f = fd[0];
    memset(&f, 0, sizeof f);
    f.mult = (float)(c->cfg.max_depth / 4228250625.0);
    f.scale = (float)depth_scale;
    f.Kd[0] = K[0]; f.Kd[1] = K[4]; f.Kd[2] = K[2]; f.Kd[3] = K[5];
    f.rKd[0] = 1.0 / K[0]; f.rKd[1] = 1.0 / K[4];
    const FrameDev* dfp = nullptr;
    ParamSlot* slot = nullptr;
    int rc = stage_params(c, fd, s, &dfp, &slot);
    if (rc != MDVT_OK) return rc;
    if (d_unused) MDVT_HIP(c, hipMemsetAsync(d_unused, 0, (size_t)c->W * c->H, s));
    MDVT_HIP(c, launch_edge_filter(d_depth_rgb, depth_pitch, 0, dfp, 0, 1, c->W, c->H, of_by_one ? 1 : 0,
                                   d_tri_invalid, 0, d_unused, 0, s));
    MDVT_HIP(c, hipEventRecord(slot->done, s));
    return MDVT_OK;
}

static int ensure_ni_workspace(mdvt_ctx* c, int chunk, hipStream_t s);      // (the listed-pixel stages' workspace, below)

int mdvt_infill_using_normals(mdvt_ctx* c, const uint8_t* d_color, size_t color_pitch, const uint8_t* d_hole,
                              size_t hole_pitch, const float* d_normal, size_t normal_pitch, uint8_t* d_out,
                              size_t out_pitch, int max_steps, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_color || !d_hole || !d_normal || !d_out) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (d_out == d_color) return fail(c, MDVT_ERR_INVALID_ARG, "d_out may not alias d_color (sources are read from the input image)");
    if (color_pitch < (size_t)3 * c->W || out_pitch < (size_t)3 * c->W || hole_pitch < (size_t)c->W ||
        normal_pitch < (size_t)12 * c->W || normal_pitch % 4 != 0)
        return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (max_steps < 0) return fail(c, MDVT_ERR_INVALID_ARG, "max_steps must be >= 0");
    if (hole_pitch >= (1u << 24) || (unsigned long long)hole_pitch * c->H > 0xFFFFFFFFull)
        return fail(c, MDVT_ERR_UNSUPPORTED, "hole plane too large for the march's 32-bit offsets (pitch %zu, %d rows)", hole_pitch, c->H);
    DeviceGuard g(c->device);
    if (int rc = ensure_ni_workspace(c, 1, (hipStream_t)stream)) return rc;
    MDVT_HIP(c, launch_infill_normals(d_color, color_pitch, d_hole, hole_pitch, d_normal, normal_pitch, d_out, out_pitch,
                                      c->W, c->H, max_steps, c->ni_ws, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_mark_lower_side(mdvt_ctx* c, const uint8_t* d_normals_img, size_t img_pitch, uint8_t* d_out, size_t out_pitch,
                         int max_steps, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_normals_img || !d_out) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (d_out == d_normals_img) return fail(c, MDVT_ERR_INVALID_ARG, "d_out may not alias the input image");
    if (img_pitch < (size_t)3 * c->W || out_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (max_steps < 0) return fail(c, MDVT_ERR_INVALID_ARG, "max_steps must be >= 0");
    if (img_pitch >= (1u << 24) || (unsigned long long)img_pitch * c->H > 0xFFFFFFFFull)
        return fail(c, MDVT_ERR_UNSUPPORTED, "image too large for the march's 32-bit offsets (pitch %zu, %d rows)", img_pitch, c->H);
    DeviceGuard g(c->device);
    if (int rc = ensure_ni_workspace(c, 1, (hipStream_t)stream)) return rc;
    MDVT_HIP(c, launch_mark_lower_side(d_normals_img, img_pitch, d_out, out_pitch, c->W, c->H, max_steps, c->ni_ws, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_touchly_depth(mdvt_ctx* c, const float* d_depth, size_t depth_pitch, uint8_t* d_rgb, size_t rgb_pitch,
                       double touchly_max_depth, double touchly_min_depth, int zero_is_far, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_depth || !d_rgb) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (depth_pitch < (size_t)4 * c->W || rgb_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (!(touchly_max_depth > touchly_min_depth)) return fail(c, MDVT_ERR_INVALID_ARG, "touchly_max_depth must exceed touchly_min_depth");
    DeviceGuard g(c->device);
    // NumPy: f32 array (op) python float -> the scalar is rounded to f32 first
    MDVT_HIP(c, launch_touchly_depth(d_depth, depth_pitch, d_rgb, rgb_pitch, c->W, c->H, (float)touchly_max_depth,
                                     (float)touchly_min_depth, (float)(255.0 / (touchly_max_depth - touchly_min_depth)),
                                     zero_is_far, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_equirect_tables(int width, int height, double input_fov_deg, float* h_map_x, float* h_map_y)
{
    if (width < 2 || height < 2 || !h_map_x || !h_map_y) return MDVT_ERR_INVALID_ARG;
    if (!(input_fov_deg > 0.0 && input_fov_deg < 180.0)) return MDVT_ERR_INVALID_ARG;
    // f64 throughout, rounded to f32 at the end like map_x.astype(np.float32) (sr:77-78)
    const double pi = 3.141592653589793;
    const double cx = ((double)width - 1.0) / 2.0, cy = ((double)height - 1.0) / 2.0;
    const double half = (input_fov_deg / 2.0) * (pi / 180.0);
    const double fx = cx / tan(half), fy = cy / tan(half);
    for (int x = 0; x < width; ++x) {
        const double theta = ((double)x - cx) / cx * (pi / 2.0);
        h_map_x[x] = fabs(theta) <= half ? (float)(fx * tan(theta) + cx) : -1.0f;
    }
    for (int y = 0; y < height; ++y) {
        const double phi = ((double)y - cy) / cy * (pi / 2.0);
        h_map_y[y] = fabs(phi) <= half ? (float)(fy * tan(phi) + cy) : -1.0f;
    }
    return MDVT_OK;
}

int mdvt_equirect_remap(mdvt_ctx* c, const uint8_t* d_src, size_t src_pitch, size_t src_stride, uint8_t* d_dst,
                        size_t dst_pitch, size_t dst_stride, int n_images, const float* d_map_x, const float* d_map_y,
                        void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_src || !d_dst || !d_map_x || !d_map_y) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (n_images < 1) return fail(c, MDVT_ERR_INVALID_ARG, "n_images must be >= 1");
    if (src_pitch < (size_t)3 * c->W || dst_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (d_src == d_dst) return fail(c, MDVT_ERR_INVALID_ARG, "d_dst may not alias d_src");
    DeviceGuard g(c->device);
    MDVT_HIP(c, launch_equirect_remap(d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_images, c->W, c->H,
                                      d_map_x, d_map_y, (hipStream_t)stream));
    return MDVT_OK;
}

namespace {
// cv2.getGaussianKernel(6, 0) as published: sigma = 0.3*((n-1)*0.5 - 1) + 0.8, exp in f64, scaled by 1/sum; the
// 2-D kernel is the f64 outer product (sr:124-125) rounded to f32 (what filter2D does for an f32 image).
mdvt::BlurKernel masked_blur_kernel()
{
    double g[6], sum = 0.0;
    const double sigma = 0.3 * ((6 - 1) * 0.5 - 1.0) + 0.8, scale2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < 6; ++i) { const double x = (double)i - (6 - 1) * 0.5; g[i] = exp(scale2 * x * x); sum += g[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < 6; ++i) g[i] *= sum;
    mdvt::BlurKernel K;
    for (int y = 0; y < 6; ++y) for (int x = 0; x < 6; ++x) K.k[6 * y + x] = (float)(g[y] * g[x]);
    return K;
}
constexpr int kTeleaChunk = mdvt::kTeleaMaxImages;      // images per pass (14 B/px of workspace each)

}  // namespace

int mdvt_swap_rb(mdvt_ctx* c, const uint8_t* d_src, size_t src_pitch, size_t src_stride, uint8_t* d_dst, size_t dst_pitch,
                 size_t dst_stride, int n_images, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_src || !d_dst) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (n_images < 1) return fail(c, MDVT_ERR_INVALID_ARG, "n_images must be >= 1");
    if (src_pitch < (size_t)3 * c->W || dst_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    DeviceGuard g(c->device);
    const mdvt::ImageSet in{const_cast<uint8_t*>(d_src), src_pitch, src_stride, 0, n_images}, out{d_dst, dst_pitch, dst_stride, 0, n_images};
    MDVT_HIP(c, launch_swap_rb(in, out, n_images, c->W, c->H, (hipStream_t)stream));
    return MDVT_OK;
}

int mdvt_masked_blur(mdvt_ctx* c, const uint8_t* d_img, size_t img_pitch, uint8_t* d_out, size_t out_pitch, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_img || !d_out) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (img_pitch < (size_t)3 * c->W || out_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (d_img == d_out) return fail(c, MDVT_ERR_INVALID_ARG, "d_out may not alias d_img");
    DeviceGuard g(c->device);
    const mdvt::ImageSet in{const_cast<uint8_t*>(d_img), img_pitch, 0, 0, 1}, out{d_out, out_pitch, 0, 0, 1};
    MDVT_HIP(c, launch_masked_blur(in, nullptr, out, 1, c->W, c->H, masked_blur_kernel(), 0u, (hipStream_t)stream));
    return MDVT_OK;
}

constexpr int kNormalInfillChunk = 16;       // images per launch set

static int ensure_ni_workspace(mdvt_ctx* c, int chunk, hipStream_t s)
{
    if (c->ni_images >= chunk) return MDVT_OK;
    MDVT_HIP(c, hipDeviceSynchronize());                 // earlier submissions may still use the old workspace
    if (c->ni_ws) ws_free(c, c->ni_ws);
    c->ni_ws = nullptr; c->ni_images = 0;
    MDVT_HIP(c, ws_malloc(c, (void**)&c->ni_ws, mdvt::normal_infill_workspace_bytes(chunk, c->W, c->H), s));
    c->ni_images = chunk;
    return MDVT_OK;
}


int mdvt_normal_infill(mdvt_ctx* c, const uint8_t* d_img, size_t img_pitch, size_t img_stride, const uint8_t* d_infill_mask,
                       size_t mask_pitch, size_t mask_stride, uint8_t* d_out, size_t out_pitch, size_t out_stride, int n_images,
                       void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_img || !d_infill_mask || !d_out) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (n_images < 1) return fail(c, MDVT_ERR_INVALID_ARG, "n_images must be >= 1");
    if (img_pitch < (size_t)3 * c->W || mask_pitch < (size_t)3 * c->W || out_pitch < (size_t)3 * c->W)
        return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (d_out == d_img || d_out == d_infill_mask) return fail(c, MDVT_ERR_INVALID_ARG, "d_out may not alias an input");
    if (mask_pitch >= (1u << 24) || (unsigned long long)mask_pitch * c->H > 0xFFFFFFFFull || (unsigned long long)c->W * c->H > 0xFFFFFFFFull)
        return fail(c, MDVT_ERR_UNSUPPORTED, "image too large for the marches' 32-bit offsets (pitch %zu, %d x %d)", mask_pitch, c->W, c->H);
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int chunk = n_images < kNormalInfillChunk ? n_images : kNormalInfillChunk;
    if (int rc = ensure_ni_workspace(c, chunk, s)) return rc;
    const mdvt::BlurKernel K = masked_blur_kernel();
    for (int i0 = 0; i0 < n_images; i0 += chunk) {
        const int n = n_images - i0 < chunk ? n_images - i0 : chunk;
        const mdvt::ImageSet img{const_cast<uint8_t*>(d_img) + (size_t)i0 * img_stride, img_pitch, img_stride, 0, n};
        const mdvt::ImageSet mask{const_cast<uint8_t*>(d_infill_mask) + (size_t)i0 * mask_stride, mask_pitch, mask_stride, 0, n};
        const mdvt::ImageSet out{d_out + (size_t)i0 * out_stride, out_pitch, out_stride, 0, n};
        MDVT_HIP(c, launch_normal_infill(img, mask, out, c->ni_ws, n, c->W, c->H, K, s));
    }
    return MDVT_OK;
}

int mdvt_infill_using_mask_normals(mdvt_ctx* c, uint8_t* d_img, size_t img_pitch, size_t img_stride, const uint8_t* d_hole,
                                   size_t hole_pitch, size_t hole_stride, const uint8_t* d_mask_img, size_t mask_pitch,
                                   size_t mask_stride, int n_images, int max_steps, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_img || !d_hole || !d_mask_img) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if (n_images < 1) return fail(c, MDVT_ERR_INVALID_ARG, "n_images must be >= 1");
    if (img_pitch < (size_t)3 * c->W || mask_pitch < (size_t)3 * c->W || hole_pitch < (size_t)c->W)
        return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (max_steps < 0) return fail(c, MDVT_ERR_INVALID_ARG, "max_steps must be >= 0");
    if (d_img == d_mask_img) return fail(c, MDVT_ERR_INVALID_ARG, "d_img may not alias d_mask_img");
    if (hole_pitch >= (1u << 24) || (unsigned long long)hole_pitch * c->H > 0xFFFFFFFFull)
        return fail(c, MDVT_ERR_UNSUPPORTED, "hole plane too large for the march's 32-bit offsets (pitch %zu, %d rows)", hole_pitch, c->H);
    DeviceGuard g(c->device);
    const int chunk = n_images < kNormalInfillChunk ? n_images : kNormalInfillChunk;
    if (int rc = ensure_ni_workspace(c, chunk, (hipStream_t)stream)) return rc;
    for (int i0 = 0; i0 < n_images; i0 += chunk) {
        const int n = n_images - i0 < chunk ? n_images - i0 : chunk;
        const mdvt::ImageSet img{d_img + (size_t)i0 * img_stride, img_pitch, img_stride, 0, n};
        const mdvt::ImageSet hole{const_cast<uint8_t*>(d_hole) + (size_t)i0 * hole_stride, hole_pitch, hole_stride, 0, n};
        const mdvt::ImageSet mask{const_cast<uint8_t*>(d_mask_img) + (size_t)i0 * mask_stride, mask_pitch, mask_stride, 0, n};
        MDVT_HIP(c, launch_infill_mask_normals(img, hole, mask, c->ni_ws, n, c->W, c->H, max_steps, (hipStream_t)stream));
    }
    return MDVT_OK;
}

static int finish_infill_mask(mdvt_ctx* c, const uint8_t* d_seed, const uint8_t* d_seed_right, size_t seed_pitch, size_t seed_stride,
                              uint8_t* d_out, uint8_t* d_out_right, size_t out_pitch, size_t out_stride, int n_frames, int max_rounds,
                              uint32_t* d_remaining, void* stream)
{
    if (!c) return MDVT_ERR_INVALID_ARG;
    if (!d_seed || !d_out) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    if ((d_seed_right == nullptr) != (d_out_right == nullptr)) return fail(c, MDVT_ERR_INVALID_ARG, "right-eye seed and output go together");
    if (n_frames < 1) return fail(c, MDVT_ERR_INVALID_ARG, "n_images must be >= 1");
    if (seed_pitch < (size_t)3 * c->W || out_pitch < (size_t)3 * c->W) return fail(c, MDVT_ERR_INVALID_ARG, "pitch smaller than one row");
    if (d_seed == d_out || (d_seed_right && d_seed_right == d_out_right)) return fail(c, MDVT_ERR_INVALID_ARG, "d_out may not alias d_seed");
    // max_rounds < 0: |max_rounds| levels, every one of them launched without asking the device how many exist (no wait on the stream)
    const bool no_wait = max_rounds < 0;
    if (no_wait) max_rounds = -max_rounds;
    if (max_rounds == 0) max_rounds = 256;
    if (max_rounds > 32766) return fail(c, MDVT_ERR_INVALID_ARG, "max_rounds must be <= 32766");
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int W = c->W, H = c->H;
    const size_t npx = (size_t)W * H;
    if ((unsigned long long)kTeleaChunk * npx > 0xFFFFFFFFull)      // work-list entries are 32-bit pixel indices over a full pass
        return fail(c, MDVT_ERR_UNSUPPORTED, "frame too large for the infill-mask completion (%d x %d)", W, H);
    const int eyes_ = d_seed_right ? 2 : 1;
    const int want_images = n_frames * eyes_ < kTeleaChunk ? n_frames * eyes_ : kTeleaChunk;
    if (c->telea_images < want_images || c->telea_rounds < max_rounds) {
        MDVT_HIP(c, hipDeviceSynchronize());                 // earlier submissions may still use the old workspace
        const int images = want_images > c->telea_images ? want_images : c->telea_images;   // per-pixel arrays; the counters hold a full pass
        const int rounds = max_rounds > c->telea_rounds ? max_rounds : c->telea_rounds;
        free_telea(c);
        mdvt::TeleaWorkspace& w = c->telea;
        MDVT_HIP(c, ws_malloc(c, (void**)&w.stamp, (size_t)images * npx * sizeof(uint16_t), s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.T, (size_t)images * npx * sizeof(float), s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.img, (size_t)images * npx * 3 + 4, s));      // + 4: pixels are fetched as unaligned dwords
        MDVT_HIP(c, ws_malloc(c, (void**)&w.need, (size_t)images * npx, s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.nlist, (size_t)images * npx * sizeof(uint32_t), s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.counts, mdvt::telea_counter_words(rounds) * sizeof(uint32_t), s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.remaining, (size_t)kTeleaChunk * sizeof(uint32_t), s));
        MDVT_HIP(c, ws_malloc(c, (void**)&w.last_round, (size_t)kTeleaChunk * sizeof(uint32_t), s));
        c->telea_images = images; c->telea_rounds = rounds;
    }
    c->telea.offs = c->telea.counts + (max_rounds + 2);
    c->telea.ncounts = c->telea.offs + (max_rounds + 2);
    if (!c->telea_levels_host) {
        void *h = nullptr, *d = nullptr;
        size_t got = 0;
        MDVT_HIP(c, pool_take(64, false, -1, &h, &d, &got));          // pinned, from the process-wide pool (see pool_take)
        c->telea_levels_host = (uint32_t*)h;
    }
    const uint32_t key = (uint32_t)c->cfg.key_rgb[0] | ((uint32_t)c->cfg.key_rgb[1] << 8) | ((uint32_t)c->cfg.key_rgb[2] << 16);
    const mdvt::BlurKernel K = masked_blur_kernel();
    const int eyes = d_seed_right ? 2 : 1;
    const int fchunk = kTeleaChunk / eyes;                   // frames per pass: both eyes of a frame travel together
    for (int f0 = 0; f0 < n_frames; f0 += fchunk) {
        const int nf = n_frames - f0 < fchunk ? n_frames - f0 : fchunk, n = nf * eyes;
        const mdvt::ImageSet seed{const_cast<uint8_t*>(d_seed) + (size_t)f0 * seed_stride, seed_pitch, seed_stride,
                                  d_seed_right ? d_seed_right - d_seed : 0, nf};
        const mdvt::ImageSet out{d_out + (size_t)f0 * out_stride, out_pitch, out_stride, d_out_right ? d_out_right - d_out : 0, nf};
        const mdvt::ImageSet work{c->telea.img, (size_t)3 * W, 3 * npx, 0, n};
        MDVT_HIP(c, launch_telea_init(seed, c->telea, n, W, H, max_rounds, key, no_wait ? nullptr : c->telea_levels_host, s));
        MDVT_HIP(c, launch_telea_rounds(c->telea, W, H, no_wait ? max_rounds : (int)*c->telea_levels_host, key, s));               // sr:806, inpaintRadius = 3
        // (the level lists and T are done with: their storage serves the blur's per-row pixel lists and row counters)
        MDVT_HIP(c, launch_masked_blur(work, &seed, out, n, W, H, K, key, s, c->telea.nlist, reinterpret_cast<uint32_t*>(c->telea.T)));   // sr:807-808
        if (d_remaining) {      // image order of the result: left eyes of all frames, then right eyes
            for (int e = 0; e < eyes; ++e)
                MDVT_HIP(c, hipMemcpyAsync(d_remaining + (size_t)e * n_frames + f0, c->telea.remaining + (size_t)e * nf,
                                           (size_t)nf * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        }
    }
    return MDVT_OK;
}

int mdvt_finish_infill_mask(mdvt_ctx* c, const uint8_t* d_seed, size_t seed_pitch, size_t seed_stride, uint8_t* d_out,
                            size_t out_pitch, size_t out_stride, int n_images, int max_rounds, uint32_t* d_remaining, void* stream)
{
    return finish_infill_mask(c, d_seed, nullptr, seed_pitch, seed_stride, d_out, nullptr, out_pitch, out_stride, n_images, max_rounds,
                              d_remaining, stream);
}

int mdvt_finish_infill_mask_stereo(mdvt_ctx* c, const uint8_t* d_left_seed, const uint8_t* d_right_seed, size_t seed_pitch,
                                   size_t seed_stride, uint8_t* d_left_out, uint8_t* d_right_out, size_t out_pitch, size_t out_stride,
                                   int n_frames, int max_rounds, uint32_t* d_remaining, void* stream)
{
    if (c && (!d_right_seed || !d_right_out)) return fail(c, MDVT_ERR_INVALID_ARG, "NULL buffer");
    return finish_infill_mask(c, d_left_seed, d_right_seed, seed_pitch, seed_stride, d_left_out, d_right_out, out_pitch, out_stride,
                              n_frames, max_rounds, d_remaining, stream);
}

}  // extern "C"
