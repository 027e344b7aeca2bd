/*
 * mdvt.h -- C ABI of the MI355X-native stereo-rerender path (libmdvt_hip.so).
 *
 * The reference (calledit/metric_depth_video_toolbox) is pure Python and exposes no FFI: the
 * path's interface is the frame loop of stereo_rerender.py (sr:471-944) and the helper functions
 * it calls.  Each entry point below names the reference code it replaces (file:line into the
 * reference tree; dfh = depth_frames_helper.py, dmt = depth_map_tools.py, sr = stereo_rerender.py).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs.  No torch / HIP types in any signature (a hipStream_t is
 *     passed as void*; NULL = the default stream).
 *   - every image pointer is DEVICE memory owned by the caller (e.g. torch.Tensor.data_ptr());
 *     the library never frees or retains it beyond the stream-ordered work of the call.
 *   - calls are asynchronous w.r.t. the host on the given stream.
 *   - return 0 (MDVT_OK) or a negative mdvt_status; mdvt_last_error() gives the text.
 *   - one ctx per (device, stream) user; a ctx is not thread-safe, distinct ctxs are independent.  The entry points that use
 *     a library-owned workspace (render, edge filter, infill-mask completion, normal_infill, infill_using_[mask_]normals,
 *     mark_lower_side) must be issued to ONE stream per ctx at a time: two calls on one ctx in two streams would share
 *     the workspace unordered, and growing a workspace synchronises the device.
 *     (mdvt_render_stereo_batch may itself run part of a long posed / converged mesh batch on a second, library-owned stream;
 *     it forks from and joins the given stream with events, so the call's results are still ordered on the given stream.)
 *   - there is NO CPU fallback: without a HIP device mdvt_create fails with MDVT_ERR_NO_DEVICE.
 */
#ifndef MDVT_H
#define MDVT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDVT_VERSION_MAJOR 0
#define MDVT_VERSION_MINOR 15
#define MDVT_VERSION ((MDVT_VERSION_MAJOR << 16) | MDVT_VERSION_MINOR)

typedef struct mdvt_ctx mdvt_ctx;

typedef enum mdvt_status {
    MDVT_OK = 0,
    MDVT_ERR_INVALID_ARG = -1,   /* ValueError / AssertionError in the reference (sr:319-323, 507) */
    MDVT_ERR_HIP = -2,           /* a HIP runtime call failed                                      */
    MDVT_ERR_UNSUPPORTED = -3,   /* valid request this build does not implement                    */
    MDVT_ERR_NO_DEVICE = -4,     /* no usable gfx950 device                                        */
    MDVT_ERR_OOM = -5
} mdvt_status;

typedef enum mdvt_mode {
    MDVT_MODE_POINTS = 0,        /* --render_as_pointcloud: GL_POINTS size 1 (sr:576-580, dmt:1086-1102, 1510) */
    MDVT_MODE_MESH = 1           /* default: grid mesh, 2 triangles per cell (dmt:1243-1254)                   */
} mdvt_mode;

/* Clip-constant settings: the argparse-derived scalars of sr:273-316 that the loop body reads. */
typedef struct mdvt_config {
    int32_t mode;                /* mdvt_mode                                                       */
    int32_t remove_edges;        /* sr:568-573 (implied by --infill_mask / --remove_edges / --do_basic_infill) */
    int32_t edge_points;         /* !--dont_place_points_in_edges (sr:589-606); needs remove_edges.  1: splat and
                                    paint into the holes (sr:813-814); 2: splat for the seed image only, holes stay
                                    black (--do_basic_infill fills them afterwards, sr:809-812)          */
    int32_t cull;                /* mesh mode: 0 = draw both faces (default), 1 = cull back faces, 2 = cull front faces.
                                    The grid's own winding (dmt:1243-1254: counter-clockwise on screen) is the front face.
                                    dmt:1507-1556 never sets Open3D's mesh_show_back_face, whose legacy default (off) most
                                    likely means GL_CULL_FACE: either behaviour can be matched once it has been observed
                                    (an OPEN parity risk for posed / converged mesh renders without edge removal: DESIGN.md section 3) */
    double ipd_m;                /* --pupillary_distance / 1000 (sr:458-459)                         */
    double max_depth;            /* --max_depth (dfh:22)                                             */
    uint8_t key_rgb[4];          /* bg_color*255: (0,0,0), or (0,255,0) with --infill_mask (sr:555-558) */
    uint32_t workspace_mib;      /* budget, in MiB, for the slots of the posed / converged MESH path only (about 64-90 B per pixel and
                                    frame in flight: it sets how many frames one launch set takes, 1 ... 16); 0 = 4096, more than
                                    1048576 is refused.  Not covered: the four slots of the posed points path, the second huge-triangle
                                    list, the infill-mask completion and normal-infill workspaces (mdvt_workspace_bytes reports all of
                                    it).  Blocks already held are kept when the budget is lowered: they return to the process-wide pool
                                    with the context (mdvt_release_cached_memory gives them back to the driver).  The field took over
                                    `reserved1` of ABI 0.11: callers built against 0.11 must zero it.                              */
    int32_t subpixel_bits;       /* the rasteriser's sub-pixel grid, log2 of the positions per pixel a vertex is snapped to before
                                    coverage is decided -- OpenGL's GL_SUBPIXEL_BITS, an implementation constant of whatever GL
                                    ran the reference (dmt:1422-1572).  0 = 8 (what desktop GPUs report); 4 = the grid of the
                                    conformant GL the fixtures tests/golden/render_gl_*.npz were rendered with (SwiftShader), so
                                    that the path can be held to them; other values are refused.  Applies to the mesh AND to
                                    points (a size-1 point is the unit square around the snapped vertex).  New in ABI 0.14
                                    together with reserved2: the struct grew from 40 to 48 bytes                                */
    int32_t reserved2;           /* must be 0 */
} mdvt_config;

/* Per-frame parameters: what sr:515-541, 563-566 and 707-721 compute before the render calls. */
typedef struct mdvt_frame_params {
    double K[9];                 /* dmt.compute_camera_matrix(xfov, yfov, W, H), row major (dmt:902) */
    double Krender[9];           /* render_cam_matrix (== K unless --vr180, sr:526-535)              */
    double depth_scale;          /* master_fov_scale_depth (sr:537-541)                              */
    double convergence_angle;    /* radians; 0 or NaN = no toe-in (sr:707-724)                       */
    double T[16];                /* transformations[frame], row-major 4x4 (sr:563-566); used iff has_T */
    int32_t has_T;
    int32_t reserved;
} mdvt_frame_params;

/* Device buffers of one call.  *_pitch = bytes between rows, *_stride = bytes between frames of a
 * batch.  RGB images are interleaved u8 (the layout cv2 hands sr:489-509 after cvtColor).  For a
 * side-by-side output (cv2.hconcat, sr:918) point right_rgb at left_rgb + 3*W with rgb_pitch = 6*W.
 * Optional outputs may be NULL. */
typedef struct mdvt_io {
    const uint8_t* depth_rgb;  size_t depth_pitch;  size_t depth_stride;   /* RGB-coded 16-bit depth (dfh:63-75) */
    const uint8_t* color_rgb;  size_t color_pitch;  size_t color_stride;   /* colour frame                       */
    uint8_t* left_rgb; uint8_t* right_rgb; size_t rgb_pitch;  size_t rgb_stride;    /* sr:819, 907           */
    uint8_t* left_mask; uint8_t* right_mask; size_t mask_pitch; size_t mask_stride; /* 255 = hole (sr:740, 854) */
    /* (ABI 0.15: both byte masks may be NULL when the packed mask below is requested and the frames are rendered by the kernel
     *  that compacts the mask itself -- points mode, pure stereo shift, no edge removal, W % 4 == 0, W <= 4096 --: 12.25
     *  instead of 14 bytes per pixel leave the chip.  Every other path answers MDVT_ERR_INVALID_ARG.) */
    float* left_depth; float* right_depth; size_t zout_pitch; size_t zout_stride;   /* optional; 0 = background (dmt:1563) */
    /* optional compacted hole mask: 1 bit per pixel, bit k of byte b = pixel 8b+k (np.packbits(mask > 0,
     * bitorder="little")), rows padded to whole dwords: maskbits_pitch >= 4*ceil(W/32) */
    uint8_t* left_maskbits; uint8_t* right_maskbits; size_t maskbits_pitch; size_t maskbits_stride;
    /* optional: hole_counts[2*frame + eye] = number of hole pixels (uint32, overwritten) */
    uint32_t* hole_counts;
    /* optional (needs remove_edges): the infill-mask SEED image of each eye, u8 RGB -- left_img_mask of
     * sr:787-803 as (x*255).astype(uint8) just before cv2.inpaint: black outside holes; in holes the key
     * colour, fixed inward normals on the image border (sr:796-799) and, at splatted edge points, the
     * removed-vertex normal carried through the eye / pose transform, (n'+1)/2 (sr:596-606, 733, 802).
     * cv2.inpaint(TELEA) + masked_blur (sr:804-808) are left to the caller. */
    uint8_t* left_seed; uint8_t* right_seed; size_t seed_pitch; size_t seed_stride;
} mdvt_io;

int mdvt_version(void);

/* Creates a context for W x H frames on HIP device `device`.  flags: reserved, pass 0. */
int mdvt_create(mdvt_ctx** out, int device, int width, int height, uint32_t flags);
int mdvt_destroy(mdvt_ctx* ctx);
/* Text of the last error on this ctx (or of the last failed mdvt_create when ctx == NULL). */
const char* mdvt_last_error(const mdvt_ctx* ctx);

int mdvt_set_config(mdvt_ctx* ctx, const mdvt_config* cfg);

/* Device self-test of the arithmetic building blocks the kernels substitute for generic expansions (they must give
 * the bits of the correctly rounded IEEE operation the decree of DESIGN.md names; dmt:1127-1128's divisions):
 *   which = 0: rcp_exact(x) against 1/x for every f32 in [2^-32, 2^32)
 *   which = 1: rcp_div_exact(a, x) against a/x for every f32 x in [2^-32, 2^32) and 16 numerators derived from `seed`
 *   which = 2: v_cvt_pk_u8_f32 (the colour conversion of the shader) against rint / clamp [0,255] / NaN -> 0 for every f32
 * h_mismatches (host, uint64) receives the number of differing results.  Synchronous. */
int mdvt_selftest(mdvt_ctx* ctx, int which, uint64_t seed, uint64_t* h_mismatches);

/* One iteration of the frame loop sr:512-907 for both eyes: decode -> (edge filter) -> unproject ->
 * pose / eye transform -> z-buffered render -> colour-key hole mask -> (edge-point splat). */
int mdvt_render_stereo(mdvt_ctx* ctx, const mdvt_frame_params* params, const mdvt_io* io, void* stream);
/* The same for n_frames independent frames in one submission (params: host array of n_frames). */
int mdvt_render_stereo_batch(mdvt_ctx* ctx, int n_frames, const mdvt_frame_params* params,
                             const mdvt_io* io, void* stream);

/* dfh.decode_rgb_depth_frame(rgb, max_depth, True) (dfh:99-103), then sr:541's `depth *= scale`. */
int mdvt_decode_depth(mdvt_ctx* ctx, const uint8_t* d_rgb, size_t rgb_pitch, float* d_depth, size_t depth_pitch,
                      double max_depth, double depth_scale, void* stream);
/* dfh.encode_depth_as_uint32 + dfh.encode_data_as_BGR(bit16=True) (dfh:5-11, 48-61; sr:930-936).
 * bgr != 0 writes B,G,R byte order (what the reference hands to cv2.VideoWriter), else R,G,B. */
int mdvt_encode_depth(mdvt_ctx* ctx, const float* d_depth, size_t depth_pitch, uint8_t* d_rgb, size_t rgb_pitch,
                      double max_depth, int bgr, void* stream);

/* dmt.get_mesh_from_depth_map(..., remove_edges=True, return_normals_of_removed=True)'s filter
 * (dmt:1283-1294, 1339-1376) on its own: d_tri_invalid[2*(H-1)*(W-1)] u8 in draw order (all tri1 then
 * all tri2), d_unused[H*W] u8 (1 = vertex of a removed triangle).  Either output may be NULL. */
int mdvt_edge_filter(mdvt_ctx* ctx, const uint8_t* d_depth_rgb, size_t depth_pitch, const double K[9],
                     double depth_scale, int of_by_one, uint8_t* d_tri_invalid, uint8_t* d_unused, void* stream);

/* Device memory the context currently owns (workspaces of the general paths, the edge filter, the infill-mask completion,
 * normal_infill; allocated on first use, kept until mdvt_destroy or until a larger request replaces them; sizes are rounded up
 * to the pool's size classes: at most 6.25 % over the request below 1 MiB, 64 KiB steps above).  The render calls themselves allocate nothing else: every image
 * buffer is the caller's. */
int mdvt_workspace_bytes(mdvt_ctx* ctx, uint64_t* bytes);

/* Workspace blocks of destroyed (or grown) contexts are kept by the process for the next context on the same GPU instead of going
 * back to the driver (a block fresh from the driver is filled and synchronised once before its first use; see DESIGN.md section 9
 * for the r04 finding behind this); beyond 4 GiB of idle blocks per GPU (mdvt_set_cached_memory_limit) that GPU's oldest are released.  mdvt_release_cached_memory returns the
 * idle blocks of GPU `device` (-1: of every GPU) to the driver -- torch.cuda.empty_cache()'s role; it synchronises the device.
 * mdvt_cached_memory reports them (either pointer may be NULL).  No reference counterpart: Open3D / NumPy own their memory. */
int mdvt_release_cached_memory(int device);
int mdvt_cached_memory(int device, uint64_t* idle_bytes, uint64_t* idle_blocks);
/* The most idle bytes kept PER GPU (accounted per GPU; default 4 GiB = the default workspace_mib budget); whatever a GPU holds
 * beyond the new limit goes back to the driver at once, oldest first (synchronises those GPUs).  0 = keep nothing: every
 * context's blocks return to the driver when it goes (the pre-r05 behaviour, with the fresh-block treatment still applied).
 * Process-wide, thread-safe.  ABI 0.15. */
int mdvt_set_cached_memory_limit(uint64_t bytes_per_gpu);
/* libmdvt_hip_tuning.so only (MDVT_ERR_UNSUPPORTED in the product library): what = 0 copies the general mesh path's triangle-queue
 * block to h_dst (host, `capacity` bytes; NULL: sizes only) after a device synchronisation.  info: bytes of the block; dword offsets
 * of the segment counters / the huge list / the tie flags; segments; W; H; frame slots.  tests/dbg_stress_case.py's diagnosis.
 * what = 1: the cross-XCD coherence test on that block (h_dst: 80 dwords; overwrites the block).  what = 2: a census of the two
 * process-wide pools as this context's GPU sees them -- info[0..3] = idle parameter blocks of this / of another GPU, idle workspace
 * blocks of this / of another GPU; info[4] = the context's pool tag (tests/test_gpu_fresh_context.py). */
int mdvt_debug_read(mdvt_ctx* ctx, int what, void* h_dst, uint64_t capacity, uint64_t info[8]);

/* Diagnostic: where the edge point of EVERY vertex of one frame lands (sr:589-606, 615-619, 727-735, 745-750, 838-858: the
 * vertices of removed triangles, undo-scaled, taken through pose / convergence / +-ipd/2, cv2.projectPoints with the f32-cast
 * camera matrix (dmt:1057-1060), np.round).  d_px: int32 [H*W][2 eyes][2] = (x, y), INT32_MIN twice where the rounded pixel
 * lies outside the frame or the vertex has depth code 0 (not splatted).  how = 0: the f64 chain of the reference's
 * operations, evaluated per vertex as the general kernels do; how = 1 (pure-shift frames only): as the LDS row kernels
 * take it -- column from an f32 estimate wherever that is provably the chain's, the chain otherwise; row = the source row
 * except on the scanlines the frame's camera matrix sends elsewhere.  Both must give the same numbers.  Uses mode, ipd_m
 * and max_depth of the config. */
int mdvt_edge_point_pixels(mdvt_ctx* ctx, const mdvt_frame_params* params, const uint8_t* d_depth_rgb, size_t depth_pitch,
                           int how, int32_t* d_px, void* stream);

/* stereo_rerender.infill_using_normals (sr:155-240; --do_basic_infill at sr:810-812, and
 * basic_nomal_infill.py:103): every hole pixel marches from its position along the XY direction of its
 * normal until it meets a non-hole pixel (at most max_steps pixels; the sample two, then one, step further
 * in is preferred) and takes that pixel's colour.  d_color / d_out: u8 RGB rows; d_hole: u8, nonzero = hole;
 * d_normal: f32 x 3 per pixel (XY = march direction; a pixel whose normal is exactly (0,1,0) or whose XY
 * length is <= 1e-6 is left alone).  d_out may not alias d_color. */
int mdvt_infill_using_normals(mdvt_ctx* ctx, const uint8_t* d_color, size_t color_pitch, const uint8_t* d_hole,
                              size_t hole_pitch, const float* d_normal, size_t normal_pitch, uint8_t* d_out,
                              size_t out_pitch, int max_steps, void* stream);

/* infill_common.mark_lower_side (infill_common.py:4-49; used by basic_nomal_infill.py:111): every non-black
 * pixel of the normal-coloured mask image marches along its encoded XY direction ((rg/255)*2-1); where it
 * first steps onto a black pixel the previous sample position is painted (0,0,255) in d_out (all other
 * pixels 0).  u8 RGB rows, max_steps as in the reference (march for t = 1 .. max_steps-1). */
int mdvt_mark_lower_side(mdvt_ctx* ctx, const uint8_t* d_normals_img, size_t img_pitch, uint8_t* d_out,
                         size_t out_pitch, int max_steps, void* stream);

/* Touchly depth plane (sr:549-551 fast path, sr:689-691 / 825-829 after a render): f32 metres -> u8 RGB with
 * all three channels = 255 - rint(max(0, min(depth, touchly_max) - touchly_min) * 255/(touchly_max-touchly_min))
 * (f32 arithmetic as NumPy evaluates it).  zero_is_far != 0 applies sr:690 / 827 first: a quantised value of 0
 * (render background) is treated as the far plane. */
int mdvt_touchly_depth(mdvt_ctx* ctx, const float* d_depth, size_t depth_pitch, uint8_t* d_rgb, size_t rgb_pitch,
                       double touchly_max_depth, double touchly_min_depth, int zero_is_far, void* stream);

/* stereo_rerender.convert_to_equirectangular (sr:25-86; --vr180 / --touchly0 at sr:914-916): the rectilinear
 * render (input_fov across the frame) placed in the centre of a 180-degree equirectangular image of the same
 * size.  The reference's two H x W float32 maps are separable, so they are passed as lookup tables:
 * map_x[W], map_y[H], -1 = angle outside the input fov (the pixel becomes black, like the (-1,-1) map entry
 * with BORDER_CONSTANT).  mdvt_equirect_tables fills HOST arrays with the reference's f64 arithmetic
 * (sr:41-78; no ctx, no device work); the caller uploads them once per (W, H, fov). */
int mdvt_equirect_tables(int width, int height, double input_fov_deg, float* h_map_x, float* h_map_y);
/* cv2.remap(image, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT, 0) (sr:82-84) for n_images u8 RGB images of the
 * ctx's W x H (stride = bytes between images): coordinates rounded to 1/32 px, integer bilinear weights with
 * sum 2^15, result (sum + 2^14) >> 15 -- OpenCV's fixed-point path.  d_map_x / d_map_y: DEVICE tables. */
int mdvt_equirect_remap(mdvt_ctx* ctx, const uint8_t* d_src, size_t src_pitch, size_t src_stride, uint8_t* d_dst,
                        size_t dst_pitch, size_t dst_stride, int n_images, const float* d_map_x, const float* d_map_y,
                        void* stream);

/* cv2.cvtColor(frame, COLOR_BGR2RGB) on the way in (sr:493, 505) and COLOR_RGB2BGR on the way out (sr:928, 941):
 * bytes 0 and 2 of every pixel of n_images interleaved u8 images of the ctx's W x H swap.  In place (d_dst == d_src with
 * equal pitch / stride) is allowed. */
int mdvt_swap_rb(mdvt_ctx* ctx, const uint8_t* d_src, size_t src_pitch, size_t src_stride, uint8_t* d_dst, size_t dst_pitch,
                 size_t dst_stride, int n_images, void* stream);

/* stereo_rerender.masked_blur(img, ksize=(6,6), sigma=0) (sr:114-153): a Gaussian that ignores pure black pixels
 * (black stays black).  cv2.getGaussianKernel / cv2.filter2D(BORDER_ISOLATED) by their published definitions: f32
 * correlation with the f64-built 6x6 kernel, anchor (3,3), zero border, 36 taps summed row-major without
 * contraction, result truncated to u8.  u8 RGB rows of the ctx's W x H. */
int mdvt_masked_blur(mdvt_ctx* ctx, const uint8_t* d_img, size_t img_pitch, uint8_t* d_out, size_t out_pitch, void* stream);

/* The completion of the infill-mask image, sr:803-808 + 816, for n_images seed images (the left_seed / right_seed
 * outputs of mdvt_render_stereo): every key-coloured or black pixel is inpainted from the normal-coloured ones
 * (cv2.inpaint(..., 3, INPAINT_TELEA)'s role), the key-coloured pixels keep the inpainted value, black ones return
 * to black, then masked_blur.  The inpaint uses Telea's weights as OpenCV publishes them but fills LEVEL BY LEVEL
 * (round r = every unknown pixel with a 4-neighbour known before round r), not in OpenCV's one-pixel-at-a-time
 * heap order -- the parallel form of the fast-marching front; it is not bit-identical to cv2.inpaint.
 * max_rounds (<= 0: 256, at most 32766) bounds the front's travel, in pixels of 4-neighbour distance from the nearest
 * seed; the rounds stop at the level of the deepest key-coloured pixel, and d_remaining (optional, n_images x uint32)
 * receives the number of key-coloured pixels beyond max_rounds.  Up to 32 images share one pass (14 B/px of workspace
 * each).  With max_rounds >= 0 this entry point -- unlike the others -- WAITS on the stream once per pass: the levels come from a
 * distance transform, and the host reads the deepest level back so that exactly that many level launches follow (the read-back
 * word is per-ctx state: like every entry point, not to be called on one ctx from two threads at once).
 * max_rounds < 0 (ABI 0.15) is the ASYNCHRONOUS form: |max_rounds| levels are launched without asking the device how many exist
 * (two launches per level; a level that does not exist finds an empty list and returns, ~3 us of device time each), nothing waits
 * on the stream: the call only enqueues work, like a render, and the caller's thread is free to stage the next batch.  Same bytes as the waiting form with the same bound; a
 * caller that knows its clips (d_remaining of earlier frames tells whether a bound was enough) passes a tight one.
 * The key colour is the ctx's cfg.key_rgb.  d_out may not alias d_seed. */
int mdvt_finish_infill_mask(mdvt_ctx* ctx, const uint8_t* d_seed, size_t seed_pitch, size_t seed_stride, uint8_t* d_out,
                            size_t out_pitch, size_t out_stride, int n_images, int max_rounds, uint32_t* d_remaining,
                            void* stream);

/* The same for both eyes of n_frames frames in one pass (the seed outputs of one mdvt_render_stereo_batch call): the
 * level launches are shared by all 2 * n_frames images, which halves their count per frame.  d_remaining (optional):
 * 2 * n_frames x uint32, the left eyes of all frames first, then the right eyes. */
int mdvt_finish_infill_mask_stereo(mdvt_ctx* ctx, const uint8_t* d_left_seed, const uint8_t* d_right_seed, size_t seed_pitch,
                                   size_t seed_stride, uint8_t* d_left_out, uint8_t* d_right_out, size_t out_pitch,
                                   size_t out_stride, int n_frames, int max_rounds, uint32_t* d_remaining, void* stream);

/* basic_nomal_infill.normal_infill (basic_nomal_infill.py:87-119, called per eye at :186 and :226): the stereo image
 * d_img and its finished infill-mask image d_infill_mask (the output of mdvt_finish_infill_mask; both u8 RGB, the mask's
 * r, g = the direction to march in, black = not a hole) -> the image with its holes filled.  In the reference's order:
 * the pixels whose mask has no zero channel go black (:88-91), masked_blur (:98), infill_using_normals along
 * ((mask/255)*2-1) with max_steps 400 (:101), cv2.blur 4x4 of the result written to those pixels (:104-107),
 * mark_lower_side(mask) with max_steps 30, grown by scipy's binary_dilation(iterations=6) (:111-115), and
 * blur_under_mask inside that band (:118, 46-85).  cv2.blur is restated from OpenCV's published box filter (anchor (2,2),
 * BORDER_REFLECT_101, cvRound), cv2.filter2D as in mdvt_masked_blur.  The reference also blackens the caller's img in
 * place (:91) and writes the blurred infill into it (:107); here d_img is read only and d_out receives the returned
 * image.  n_images images share the launches (about 16 B/px of workspace each, kept by the ctx); d_out is also the work image of the passes.  d_out may not alias an input.
 * The marches address a plane with 32-bit offsets: pitches below 2^24 bytes and pitch x height below 2^32 (MDVT_ERR_UNSUPPORTED otherwise;
 * the same holds for mdvt_infill_using_normals, mdvt_infill_using_mask_normals and mdvt_mark_lower_side). */
int mdvt_normal_infill(mdvt_ctx* ctx, const uint8_t* d_img, size_t img_pitch, size_t img_stride, const uint8_t* d_infill_mask,
                       size_t mask_pitch, size_t mask_stride, uint8_t* d_out, size_t out_pitch, size_t out_stride,
                       int n_images, void* stream);

/* sr:809-812 (--do_basic_infill) for n_images images at once: infill_using_normals(image, bg_mask, mask * 2 - 1) with the
 * normals taken straight from the finished infill-mask image d_mask_img (u8 RGB; ((v / 255) * 2) - 1 in f32, as sr:808 + 810
 * evaluate it), d_hole the u8 hole plane (non-zero = hole, e.g. the mask output of mdvt_render_stereo).  d_img is filled IN
 * PLACE: a source pixel is never a hole pixel (sr:226), so the march reads what the reference's unmodified input holds. */
int mdvt_infill_using_mask_normals(mdvt_ctx* ctx, uint8_t* d_img, size_t img_pitch, size_t img_stride, const uint8_t* d_hole,
                                   size_t hole_pitch, size_t hole_stride, const uint8_t* d_mask_img, size_t mask_pitch,
                                   size_t mask_stride, int n_images, int max_steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDVT_H */
