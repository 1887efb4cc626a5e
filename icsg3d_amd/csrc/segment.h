// Declarations for segment.hip (connected components, region statistics: /root/reference/watershed.py:52-56,153-187).
#pragma once
#include "common.h"

namespace ics {

// ints per region in the statistics table: {species (majority vote, 0 = none), voxels, sum z, sum y, sum x,
// bounding box z0, y0, x0, z1, y1, x1 (half-open)}
constexpr int kSegStatInts = 11;

size_t segment_workspace_bytes(int B, int d, int max_atoms, int nbins);
// mask / species: device uint8 [B][d][d][d].  d_R: optional device int32 [B][d][d][d] region labels (0 = background or
// dropped component).  *d_counts -> device int [B][2] {components, kept components (size > min_voxels)};
// *d_stats -> device int [B][max_atoms][kSegStatInts]; both inside `workspace`.
int launch_segment_atoms(hipStream_t st, const unsigned char* mask, const unsigned char* species, int B, int d,
                         int min_voxels, int max_atoms, int nbins, void* workspace, size_t workspace_bytes, int* d_R,
                         int** d_counts, int** d_stats);

}  // namespace ics
