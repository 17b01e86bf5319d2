/* Plain-C consumer of include/lidar4d_hip.h: proves that the header is valid C (no C++ or torch types in the boundary),
 * that every declared entry point links against liblidar4d_hip.so with the declared prototype, and that the version and
 * error calls work without a GPU.  Built and run by tests/test_host_logic.py::test_c_abi_from_plain_c (gcc). */
#include <stdio.h>

#include "lidar4d_hip.h"

typedef void (*fn_t)(void);

int main(void) {
  const fn_t entry_points[] = {
      (fn_t)&l4d_adam_step,
      (fn_t)&l4d_adam_step_ranges,
      (fn_t)&l4d_attr_gather,
      (fn_t)&l4d_attr_gather_bwd,
      (fn_t)&l4d_attr_mlp_bwd,
      (fn_t)&l4d_attr_mlp_bwd_gathered,
      (fn_t)&l4d_attr_mlp_fwd,
      (fn_t)&l4d_attr_scatter,
      (fn_t)&l4d_attr_scatter_bwd,
      (fn_t)&l4d_cast_f32_to_f16,
      (fn_t)&l4d_chamfer_bwd,
      (fn_t)&l4d_lidar_ray_batch,
      (fn_t)&l4d_lidar_losses,
      (fn_t)&l4d_glue_workspace,
      (fn_t)&l4d_ray_chamfer_grad,
      (fn_t)&l4d_scale_buffers,
      (fn_t)&l4d_flow_xt,
      (fn_t)&l4d_flow_warp,
      (fn_t)&l4d_flow_chamfer_grad,
      (fn_t)&l4d_flow_loss_finish,
      (fn_t)&l4d_flow_dy16,
      (fn_t)&l4d_axpy_dev,
      (fn_t)&l4d_chamfer_fwd,
      (fn_t)&l4d_chamfer_workspace,
      (fn_t)&l4d_composite_bwd,
      (fn_t)&l4d_composite_fwd,
      (fn_t)&l4d_composite_image,
      (fn_t)&l4d_density_encode_bwd,
      (fn_t)&l4d_density_encode_bwd_workspace,
      (fn_t)&l4d_density_encode_fwd,
      (fn_t)&l4d_density_encode_sigma_fwd,
      (fn_t)&l4d_density_encode_fwd_workspace,
      (fn_t)&l4d_dyn_pairs_build,
      (fn_t)&l4d_field_width,
      (fn_t)&l4d_freq_fwd,
      (fn_t)&l4d_grad_nonfinite_check,
      (fn_t)&l4d_absmax_f32,
      (fn_t)&l4d_hashgrid_bwd,
      (fn_t)&l4d_hashgrid_fwd,
      (fn_t)&l4d_hashgrid_fwd_workspace,
      (fn_t)&l4d_hashgrid_fwd_ws,
      (fn_t)&l4d_hashgrid_t_bwd,
      (fn_t)&l4d_hashgrid_t_bwd_workspace,
      (fn_t)&l4d_hashgrid_t_fwd,
      (fn_t)&l4d_hashgrid_t_fwd_workspace,
      (fn_t)&l4d_hashgrid_t_fwd_ws,
      (fn_t)&l4d_last_error,
      (fn_t)&l4d_lidar_to_pano,
      (fn_t)&l4d_lidar_to_pano_workspace,
      (fn_t)&l4d_mark_time_slices,
      (fn_t)&l4d_mlp_bwd,
      (fn_t)&l4d_mlp_fwd,
      (fn_t)&l4d_mlp_fwd_sigma,
      (fn_t)&l4d_pano_to_lidar,
      (fn_t)&l4d_pano_to_lidar_workspace,
      (fn_t)&l4d_planes_bwd,
      (fn_t)&l4d_planes_fwd,
      (fn_t)&l4d_plane_rows_workspace,
      (fn_t)&l4d_planes_relayout,
      (fn_t)&l4d_profile_count,
      (fn_t)&l4d_profile_enable,
      (fn_t)&l4d_profile_get,
      (fn_t)&l4d_sample_rays,
      (fn_t)&l4d_sample_rays_xt,
      (fn_t)&l4d_scaler_update,
      (fn_t)&l4d_side_fork,
      (fn_t)&l4d_side_join,
      (fn_t)&l4d_sigma_bwd,
      (fn_t)&l4d_sigma_bwd_rows,
      (fn_t)&l4d_sigma_from_h,
      (fn_t)&l4d_streams_config,
      (fn_t)&l4d_streams_join,
      (fn_t)&l4d_streams_mask,
      (fn_t)&l4d_time_setup,
      (fn_t)&l4d_version,
  };
  const int n = (int)(sizeof(entry_points) / sizeof(entry_points[0]));
  for (int i = 0; i < n; ++i)
    if (!entry_points[i]) return 2;
  if (l4d_version() != L4D_ABI_VERSION) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", l4d_version(), L4D_ABI_VERSION);
    return 3;
  }
  printf("%d entry points, ABI v%d, last error: \"%s\"\n", n, l4d_version(), l4d_last_error());
  return 0;
}
