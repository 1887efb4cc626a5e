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
                         int** d_counts, int** d_stats, long long** d_bounds = nullptr);
// d_bounds (optional): *d_bounds -> device int64 [B][max_atoms][8] inside `workspace`: {grid points of the region's bounding
// box inside its 26-direction polytope (an upper bound of the convex hull image's count), sum zz, yy, xx, zy, zx, yx of
// the voxel coordinates, 0} -- what the convexity test of segment_nuclei is decided from for most components.
constexpr int kSegBoundInts = 8;

// ---- boxes: small dense int32 volumes [D][H][W] (extents <= 64), `nbox` of them back to back; HOST pointers in and out
// (tens of KB per call: the recursion of segment_nuclei is driven from the host, watershed.py:40-150).
// skimage.measure.label(box, connectivity): components of equal non-zero value, raster-order numbering; connectivity 1
// (6 neighbours) or 3 (26).  stats (optional): [nbox][max_labels][7] = {count, z0, y0, x0, z1, y1, x1}.
int segment_label_boxes(hipStream_t st, const int* h_vols, const int* h_dims, int nbox, int connectivity, int max_labels,
                        int* h_labels, int* h_nlabels, int* h_stats);
// centroids / majority_vote (watershed.py:153-187) for an arbitrary label volume R [D][H][W] with labels 1..nlab:
// stats [nlab][kSegStatInts], rows of labels that do not occur keep voxels = 0.  HOST pointers.
int segment_region_stats(hipStream_t st, const int* h_R, const unsigned char* h_species, int D, int H, int W, int nlab,
                         int nbins, int* h_stats);
// watershed.py:95-110 per box (values {0, cl}): ball(1) erosion / dilation, markers, priority flood, wss[wss == 1] = 0.
// tie: 0 = skimage's heap order among the age-0 markers, 1 = FIFO (oracle/watershed_ref.py::watershed_flood).
int segment_watershed_split(hipStream_t st, const int* h_boxes, const int* h_dims, const int* h_cls, int nbox, int tie,
                            int* h_wss);

// The same split on host threads (no device work: the flood is a chain of dependent heap operations);
// bit-identical to the kernel.  ics_op_watershed_split takes this form unless ICSG3D_WS_DEVICE=1.
int segment_watershed_split_host(const int* h_boxes, const int* h_dims, const int* h_cls, int nbox, int tie, int* h_wss);

// Exact-integer convexity bounds per component of labelled boxes (what ics_op_label_boxes returned), host threads:
// h_bounds [nbox][max_labels][5] = {voxels, polytope count P >= hull count, axis-line fill count F <= hull count, flat,
// H = the exact hull count where hull_threshold > 0 and the two bounds leave voxels / hull >= hull_threshold open, else 0}.
int segment_component_bounds(const int* h_labels, const int* h_dims, int nbox, const int* h_nlabels, const int* h_stats,
                             int max_labels, int min_voxels, double hull_threshold, long long* h_bounds);

// frees the calling thread's grow-only scratch of the three box-level entry points above
void segment_release_scratch();

}  // namespace ics
