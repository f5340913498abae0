// Iso-surface extraction from a TSDFVolume (zero level set of the distance field), on the HOST.
// Same entry point as the reference's extract_surface (src/include/MarkAndSweepMC.hpp:8).  The reference runs
// marching cubes on the GPU; here the distance array is copied to the host and triangulated there, cube by cube in the
// reference's order (BASELINE north_star: "src/MarchingCubes stays host-side").
#ifndef TSDF_AMD_HOST_MARK_AND_SWEEPMC_INCLUDED
#define TSDF_AMD_HOST_MARK_AND_SWEEPMC_INCLUDED

#include <vector>

#include "TSDFVolume.hpp"

void extract_surface(const TSDFVolume *volume, std::vector<float3> &vertices, std::vector<int3> &triangles);

// The same marching cubes over a host distance array (x fastest, voxel centres at (i + 0.5) * voxel_size + offset):
// appends three vertices per triangle.  extract_surface is this on the volume's distances.
void tsdf_host_marching_cubes(const float *dist, unsigned X, unsigned Y, unsigned Z, const float voxel_size[3],
                              const float offset[3], std::vector<float3> &vertices);

#endif
