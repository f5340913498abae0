"""tsdf_amd -- MI355X (gfx950) implementation of the TSDF integrate / raycast / bilateral hot path
of Scoobadood/TSDF behind the reference's class surface.  See DESIGN.md and include/tsdf_amd.h."""
from ._capi import TsdfError, last_error, LIB_PATH  # noqa: F401  (import fails loudly without the HIP library)
from .api import (TSDFVolume, GPURaycaster, BilateralFilter, Camera, ICPOdometry, compute_normals_device,  # noqa: F401
                  merge_hits_device, merge_hits_normals_device, HIT_RECORD_BYTES, vertices_to_depth_device, marching_cubes, marching_cubes_table, load_block_tsdf, load_tum_directory)
