// mdvt_grid_decls.h -- what the rasterising translation units define, declared once per sub-pixel grid: included by
// mdvt_internal.h inside namespace mdvt::grid8 and again inside mdvt::grid4 (no include guard on purpose).  The translation
// units themselves are compiled with -DMDVT_SUBPIX_BITS=8 / 4 and put their definitions into mdvt::MDVT_GRID.
hipError_t launch_render(RenderPlan& plan, const RenderArgs& a, hipStream_t s);
// mdvt_mesh_band.hip: the pure-shift mesh rows as bands (no edge removal)
bool mesh_band_supported(const RenderPlan& plan, const RenderArgs& a);
hipError_t launch_mesh_band(const RenderPlan& plan, const RenderArgs& a, hipStream_t s);
// mdvt_mesh_band3.hip: the same rows with no vertex records in LDS, both eyes per pass, three workgroups per CU
bool mesh_band3_supported(const RenderPlan& plan, const RenderArgs& a);
hipError_t launch_mesh_band3(const RenderPlan& plan, const RenderArgs& a, hipStream_t s);
// mdvt_mesh_conv.hip: mesh + convergence only, z-buffer in LDS (the product default of movie_2_3D.py:433-445)
bool mesh_conv_supported(const RenderPlan& plan, const RenderArgs& a);
hipError_t launch_mesh_conv(const RenderPlan& plan, const RenderArgs& a, hipStream_t s);
// the general paths' edge-point splat into the global edge keys, and the pass that empties the written words again
hipError_t launch_edge_points_splat(const RenderArgs& a, int n, bool as_list, bool counters_zeroed, hipStream_t s);
hipError_t launch_edge_keys_reset(const RenderArgs& a, int n, hipStream_t s);
hipError_t launch_edge_point_pixels(const uint8_t* depth, size_t pitch, const FrameDev* fp, int W, int H, int of_by_one, int how,
                                    int32_t* out, hipStream_t s);
// mdvt_mesh_general.hip: the rasteriser of the general mesh path (between the vertex pass and the resolve pass)
hipError_t launch_mesh_raster_general(const RenderPlan& plan, const RenderArgs& a, hipStream_t s);
hipError_t launch_pack_mask(const RenderArgs& a, int n, hipStream_t s);
hipError_t launch_reduce_counts(const RenderArgs& a, int n, hipStream_t s);
size_t render_lds_bytes(const RenderPlan& plan, int W);
bool render_fits_lds(const RenderPlan& plan, int W);      // can the pure-shift row kernels hold a row of this width in LDS?
bool points_fused_bits_applies(const RenderPlan& plan, const RenderArgs& a);      // k_points_rows_fast with the mask compaction fused in (byte masks optional)
hipError_t launch_divcheck(float mult, float scale, float dl, uint32_t* bad, hipStream_t s);      // FrameDev.div_slot: the short division tried on every depth code
