// mdvt_internal.h -- shared between the C-ABI host code (mdvt_api.hip) and the kernels
// (mdvt_kernels.hip).  Not part of the public interface (that is include/mdvt.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "mdvt.h"

namespace mdvt {

// Tuning / ablation / test hooks.  The PRODUCT library (libmdvt_hip.so) has none: tuning_env() is `return nullptr` there
// (mdvt_tuning_off.hip) and not even the hooks' names are in the binary; its one opt-in switch, MDVT_MESH_CONV, is read
// once, in mdvt_create.  `make tuning` links the same objects with mdvt_tuning_on.hip into libmdvt_hip_tuning.so, where
// tuning_env(TUNE_X) is getenv("MDVT_X"), re-read per launch: that library is what tools/ and the tests that force a kernel
// family load (MDVT_LIB_VARIANT=tuning, _lib.py).  Some hooks change results by design (MDVT_DEBUG_SKIP bits 0-4, MDVT_NI_SKIP).
enum TuneKey { TUNE_BLUR_ONE_PASS, TUNE_DEBUG_SKIP, TUNE_EDGE_INBAND, TUNE_FORCE_GLOBAL, TUNE_LDS_PAD, TUNE_MESH_BAND, TUNE_MESH_BAND3, TUNE_MESH_CONV, TUNE_MESH_OLD, TUNE_MESH_TPB, TUNE_NI_DUMP, TUNE_NI_SKIP, TUNE_PARAM_UPLOAD, TUNE_POINTS_CFG, TUNE_POINTS_NT, TUNE_POOL_TAG, TUNE_QUEUE_DUMP, TUNE_RASTER_CONV_OFF, TUNE_TELEA_BLOCKS, TUNE_TELEA_DUMP, TUNE_WS_CHUNK, TUNE_WS_FRESH, TUNE_WS_LAYOUT, TUNE_WS_PAD, TUNE_WS_POOL, TUNE_COUNT };
const char* tuning_env(TuneKey k);
bool tuning_build();
// The ablation hooks inside the kernels (RenderArgs.debug_skip, NormalInfillArgs.debug_skip) exist in the objects of the tuning
// library only (-DMDVT_TUNING, Makefile): the product kernels are compiled with the constant 0 and carry no such test.
#ifdef MDVT_TUNING
#define MDVT_DEBUG_SKIP(a) ((a).debug_skip)
#else
#define MDVT_DEBUG_SKIP(a) 0
#endif

// Near plane of the reference's render call: ctr.set_constant_z_near(0.0001) (dmt:1520).
constexpr float kNear = 1e-4f;
// Rasteriser sub-pixel grid (GL_SUBPIXEL_BITS) and the clamp applied before snapping (DESIGN.md "Arithmetic decree").
// The rasterising translation units are compiled once per supported grid (Makefile): 8 bits, what desktop GPUs report
// and the default of mdvt_config.subpixel_bits, and 4 bits, the grid of the GL the fixtures of tests/golden/render_gl_*.npz
// were rendered with.  Everything they define lives in namespace mdvt::MDVT_GRID; mdvt_api.hip picks per context.
#ifndef MDVT_SUBPIX_BITS
#define MDVT_SUBPIX_BITS 8
#endif
#if MDVT_SUBPIX_BITS == 8
#define MDVT_GRID grid8
#elif MDVT_SUBPIX_BITS == 4
#define MDVT_GRID grid4
#else
#error "MDVT_SUBPIX_BITS must be 4 or 8"
#endif
constexpr int kSubpixBits = MDVT_SUBPIX_BITS;
constexpr int kSubpix = 1 << kSubpixBits;
constexpr float kSnapLimit = 2097152.0f;   // 2^21 px: snapped coordinates fit int32, differences too

// Per-frame constants, derived on the host in f64 and rounded once (see fill_frame_dev()).
struct FrameDev {
    float mult;          // f32(max_depth / 255^4)                          dfh:22
    float scale;         // f32(master_fov_scale_depth)                     sr:541
    float dl;            // f32(Krender.fx * ipd/2): pure-shift disparity numerator
    float fx, fy, cx, cy;        // input camera (f32)
    float fxr, fyr, cxr, cyr;    // render camera (f32)
    float sx, sy;        // (W+1)/W, (H+1)/H in f32 for the mesh grid, 1 for points   dmt:1117-1122
    float sW, sH;        // (W-1)/W, (H-1)/H in f32: the pure-shift column estimate of an edge point (edge_col_pure)
    int32_t general;     // 0: pure +-ipd/2 shift, 1: pose / convergence / K != Krender
    int32_t conv_band;   // general, but nothing except a toe-in about the y axis small enough for k_mesh_conv (mdvt_mesh_conv.hip)
    float M[2][12];      // per eye 3x4 = Translate(+-ipd/2) * Ry(-+a) * T, f32       sr:615-619, 724-725, 832-836
    double Kd[4];        // fx, fy, cx, cy in f64 for the 89-degree edge filter       dmt:1127-1128, 1283-1294
    double rKd[2];       // 1/fx, 1/fy in f64: the edge filter's screening pass only (never the exact path)
    // The edge points' f64 chain (mdvt_device.h "edge points"): the operands of the reference's own operations
    // (sr:599-600, 615-619, 727-732, 838-847; dmt:1058) -- nothing composed, nothing pre-multiplied.
    double sWd, sHd;     // (W-1)/W, (H-1)/H                                          sr:599-600
    double hd;           // ipd/2 (the translate of sr:731, 839, 846)
    double Td[16];       // the pose, row major (has_T)                               sr:615-619
    double cs[2];        // cos, sin of the convergence angle (has_conv)             sr:719-722
    int32_t has_T, has_conv;
    // pure-shift frames: an edge point of source row i lands on row i except for i in [erow_lo, erow_hi), where it lands on
    // i or i + 1 (which of the two the chain decides per point); the LDS row kernels leave the edge points of scanlines
    // erow_lo .. erow_hi to k_edge_rows_exact.  erow_wild: some row's offset is neither (a camera matrix with cy far
    // from H/2): the frame is rendered by the global-key kernels, whose edge points take the whole chain.
    int32_t erow_lo, erow_hi, erow_wild;
    // pure-shift point frames: index of this frame's (mult, scale, dl) in the context's table of division checks (RenderArgs.divcheck), -1:
    // none.  The disparity dl / z of a frame has 65535 possible operands z -- the depth codes --: k_divcheck has tried all of them, and
    // where the 4-instruction sequence (v_rcp_f32, one multiply, two fma) gave the bits of the IEEE division every time the counter is 0
    // and the row kernel takes it instead of the 10-instruction expansion.
    int32_t div_slot;
};

// Which row of grid cells covers output scanline k of a pure-shift mesh frame, and that row's snapped extent (depends on
// H and the grid scale only: one table per context, built on the host with the decree's own f32 operations).
struct RowCell { int32_t c, Yt, Yb, pad; };     // c = -1: the scanline lies below the last vertex row

struct RenderArgs {
    const uint8_t* depth; size_t depth_pitch, depth_stride;
    const uint8_t* color; size_t color_pitch, color_stride;
    uint8_t* rgb[2];  size_t rgb_pitch, rgb_stride;
    uint8_t* mask[2]; size_t mask_pitch, mask_stride;
    float* zout[2];   size_t zout_pitch, zout_stride;
    uint8_t* maskbits[2]; size_t maskbits_pitch, maskbits_stride;   // optional 1 bit/px hole mask
    uint32_t* hole_counts;       // optional [n_frames][2]
    uint8_t* seed[2]; size_t seed_pitch, seed_stride;            // optional infill-mask seed images
    uint32_t* row_counts;        // workspace [frames in launch][2][H], zeroed per launch (when hole_counts)
    const uint32_t* divcheck;    // [kDivSlots] mismatches of the short division per parameter set (FrameDev.div_slot); 0 = proven
    uint32_t* wave_counts;       // workspace [frames in launch][H][16]: the fused points kernel's hole counts per wave (left | right << 16)
    const FrameDev* fp;          // device array, one per frame of the batch
    int32_t W, H;
    int32_t frame0;              // first frame of this launch within the batch
    uint32_t key_rgb;            // R | G<<8 | B<<16
    // workspace (general path / edge filter); per frame-in-flight slices
    unsigned long long* keys[2]; // [slot][H*W] 64-bit z keys per eye (parity scheme: mdvt_device.h)
    uint32_t key_parity;         // bit s = parity of slot s in this launch
    unsigned long long* ekeys[2];// edge-point keys per eye
    uint32_t* elist;             // the edge-key words written since the last resolve, one segment of 2 W entries (eye << 31 | pixel) per
    uint32_t* elist_count;       //   (slot, source row) and its counter: k_edge_keys_reset empties exactly those words
    uint32_t* vlist;             // mesh, general path: the vertices of removed triangles of a slot's frame as a list (source row << 16 | column),
    uint32_t* vlist_count;       //   [slot][H*W] entries + [slot] counters: what k_edge_points_splat_list runs the f64 chain over
    unsigned long long* cbuf[2]; // general mesh path: per-eye side words of the pixels with an exact depth tie between colours (draw id << 32 | rgb,
                                 //   minimum over the fragments at the winning depth); touched at those pixels only
    uint32_t* tie_flag;          // [slot]: a pixel of the slot's frame was marked as tied in this use (zeroed per launch set)
    uint32_t* tie_tiles;         // [slot][2][tie_words]: bit per kTieTile x kTieTile tile of an eye holding a marked pixel (zeroed per launch set)
    int32_t tie_words, tie_tiles_x;
    uint8_t* tri_invalid;        // [slot][2*(H-1)*(W-1)]
    uint8_t* unused;             // [slot][H*W]
    // general mesh path: queue of the triangles that are not small (kBigRecDwords dwords each), rasterised by k_mesh_raster_queue
    uint32_t* bigq; uint32_t* bigq_count; uint32_t bigq_cap;
    uint32_t* bigq_coarse;       // [(segments >> bigq_shift) + 1] queued triangles per block of 2^bigq_shift segments (behind the segment counters;
    int32_t bigq_shift;          //   set by launch_mesh_raster_general): the queue walk's first search level, summed up in LDS
    uint32_t* hugeq;             // [kHugeCap] x 2 dwords + counter + overflow slack: row blocks of the queued triangles too large for 16 lanes (k_mesh_raster_huge)
    size_t ws_stride_px;         // H*W (elements) between slots
    size_t ws_stride_tri;        // 2*(H-1)*(W-1)
    int32_t edge_paint;          // 1: edge points are painted into the holes (sr:813-814); 0: seed image only (--do_basic_infill, sr:809-812)
    const RowCell* rowcell;      // [H], mesh mode
    int32_t cull;                // 0 none, 1 back faces, 2 front faces (mdvt_config.cull)
    int32_t debug_skip;          // ablation hook for tools/kbench.py (env MDVT_DEBUG_SKIP): 1 raster, 2 resolve, 4 staging
};

// launchers implemented in mdvt_kernels.hip; all return hipError_t from hipGetLastError()
hipError_t launch_decode_depth(const uint8_t* rgb, size_t rgb_pitch, float* out, size_t out_pitch, int W, int H,
                               float mult, float scale, hipStream_t s);
hipError_t launch_encode_depth(const float* depth, size_t depth_pitch, uint8_t* rgb, size_t rgb_pitch, int W, int H,
                               double max_depth, int bgr, hipStream_t s);
hipError_t launch_zero_bytes(void* p, size_t bytes, hipStream_t s);
hipError_t launch_edge_filter(const uint8_t* depth_rgb, size_t pitch, size_t stride, const FrameDev* fp, int frame0,
                              int n, int W, int H, int of_by_one, uint8_t* tri_invalid, size_t tri_stride,
                              uint8_t* unused, size_t unused_stride, hipStream_t s);

constexpr int kDivSlots = 256;        // parameter sets (mult, scale, dl) a context keeps division checks for; later ones take the IEEE division
constexpr int kTieTile = 32;          // pixels: side of the tiles whose "holds a pixel marked as tied" bits gate the second rasteriser pass of the general mesh path
inline size_t tie_words_of(int W, int H) { return ((size_t)((W + kTieTile - 1) / kTieTile) * (size_t)((H + kTieTile - 1) / kTieTile) + 31) / 32; }
constexpr int kHugeCap = 1 << 17;     // row-block entries of huge triangles per launch set (overflow: the queue kernel keeps the triangle)
constexpr int kBigRecDwords = 2;     // a queued triangle: draw id, frame slot << 1 | eye

struct RenderPlan {
    int mode;            // mdvt_mode
    int remove_edges;
    int edge_points;
    int general;         // any frame of the launch needs the general path
    int conv;            // mesh: every frame of the launch is convergence-only (FrameDev.conv_band): k_mesh_conv instead of the global-key kernels
    int conv_raster;     // general mesh path, every frame of the launch convergence-only: k_mesh_raster_conv instead of k_mesh_raster_small
    int vec4;            // W%4==0 and every pointer/pitch 4-byte aligned
    int fused_bits;      // set by launch_render when the render kernel itself produced maskbits / hole_counts
    int n;               // frames in this launch
    int allow_conv;      // MDVT_MESH_CONV=1 when the context was created (mdvt_create): k_mesh_conv may take convergence-only frames
    int edge_rows_max;   // pure-shift launches with edge points: the most scanlines any frame leaves to k_edge_rows_exact (0: none)
    hipEvent_t after_vertices;   // general paths: recorded on the launch's stream behind the first pass (the mesh's cell walk, the points' splat; nullptr: none)
};
// The rasterising translation units (mdvt_kernels.hip's render sections, mdvt_mesh_*.hip) are compiled once per sub-pixel grid;
// what they define is declared in mdvt_grid_decls.h, once per grid namespace (below, after the shared declarations).
// (mdvt_normal_infill.hip; workspace: normal_infill_workspace_bytes(1, W, H))
hipError_t launch_infill_normals(const uint8_t* color, size_t color_pitch, const uint8_t* hole, size_t hole_pitch,
                                 const float* normal, size_t normal_pitch, uint8_t* out, size_t out_pitch, int W, int H,
                                 int max_steps, uint8_t* workspace, hipStream_t s);
hipError_t launch_mark_lower_side(const uint8_t* img, size_t img_pitch, uint8_t* out, size_t out_pitch, int W, int H,
                                  int max_steps, uint8_t* workspace, hipStream_t s);
hipError_t launch_touchly_depth(const float* depth, size_t depth_pitch, uint8_t* rgb, size_t rgb_pitch, int W, int H,
                                float tmax, float tmin, float k, int zero_is_far, hipStream_t s);
hipError_t launch_equirect_remap(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                                 size_t dst_stride, int n, int W, int H, const float* mx, const float* my, hipStream_t s);
// n images addressed as `per_eye` frames x (1 or 2) eyes: image im = frame (im % per_eye) of eye (im / per_eye).
struct ImageSet {
    uint8_t* base; size_t pitch, stride; ptrdiff_t eye_offset; int per_eye;
    __host__ __device__ uint8_t* image(int im) const { return base + (size_t)(im % per_eye) * stride + (ptrdiff_t)(im / per_eye) * eye_offset; }
};

struct TeleaWorkspace {              // per image pixel: stamp u16, T f32, work image u8x3, need u8, nlist u32 (14 B)
    uint16_t* stamp; float* T; uint8_t* img; uint8_t* need; uint32_t* nlist;
    uint32_t* counts;                // [max_rounds + 2], followed by
    uint32_t* offs;                  // [max_rounds + 2] and
    uint32_t* ncounts;               // [max_rounds + 2] x kNcStride words, one counter per cache line (one allocation: telea_counter_words)
    uint32_t* remaining;             // [images]
    uint32_t* last_round;            // [images]
};
constexpr int kTeleaMaxImages = 32;  // images per pass
size_t telea_counter_words(int max_rounds);      // counts + offs + the strided ncounts
hipError_t launch_telea_init(const ImageSet& seed, const TeleaWorkspace& ws, int n, int W, int H, int max_rounds, uint32_t key_rgb,
                             uint32_t* h_levels, hipStream_t s);
hipError_t launch_telea_rounds(const TeleaWorkspace& ws, int W, int H, int levels, uint32_t key_rgb, hipStream_t s);
hipError_t launch_swap_rb(const ImageSet& src, const ImageSet& dst, int n, int W, int H, hipStream_t s);
struct BlurKernel { float k[36]; };      // masked_blur's 6x6 Gaussian, f32, row major (built on the host in f64)
hipError_t launch_masked_blur(const ImageSet& img, const ImageSet* seed, const ImageSet& out, int n, int W, int H,
                              const BlurKernel& K, uint32_t key_rgb, hipStream_t s, uint32_t* list = nullptr, uint32_t* count = nullptr);
// mdvt_normal_infill.hip: basic_nomal_infill.normal_infill for n images (workspace: normal_infill_workspace_bytes)
size_t normal_infill_workspace_bytes(int n, int W, int H);
hipError_t launch_normal_infill(const ImageSet& img, const ImageSet& mask, const ImageSet& out, uint8_t* workspace, int n, int W, int H,
                                const BlurKernel& K, hipStream_t s);
hipError_t launch_infill_mask_normals(const ImageSet& img, const ImageSet& hole, const ImageSet& mask, uint8_t* workspace, int n, int W, int H,
                                      int max_steps, hipStream_t s);
hipError_t launch_selftest(int which, unsigned long long seed, unsigned long long* d_mism, hipStream_t s);
hipError_t launch_coherence_test(uint32_t* blk, size_t dwords, uint32_t tag, uint32_t* d_xcc, uint32_t* d_out, hipStream_t s);     // (mdvt_selftest.hip; r05 diagnosis)

namespace grid8 {
#include "mdvt_grid_decls.h"
}
namespace grid4 {
#include "mdvt_grid_decls.h"
}

}  // namespace mdvt
