/* ORACLE — test infrastructure only. Decoder-side verification stage: occupancy-masked reprojection of the geometry maps to points,
 * colour fetch, point-to-point (D1) metric. See pcc_recon.c for the reference lines it restates. */
#ifndef ORACLE_PCC_RECON_H
#define ORACLE_PCC_RECON_H
#include <stddef.h>
#include <stdint.h>

typedef struct {            /* the fields of PCCPatch the reconstruction reads (PCCPatch.h) */
  int32_t u0, v0, size_u0, size_v0;      /* position / size in the atlas, in occupancy-resolution blocks */
  int32_t u1, v1, d1;                    /* 3D offset along the tangent, bitangent and normal axis */
  int32_t normal_axis, tangent_axis, bitangent_axis;
  int32_t projection_mode;               /* 0: depth + d1, 1: d1 - depth */
  int32_t orientation;                   /* PATCH_ORIENTATION_* (PCCCommon.h:128-137) */
  int32_t lod_x, lod_y;
} oracle_patch;
typedef struct {
  int32_t width, height;                 /* atlas frame size */
  int32_t occupancy_resolution;          /* 16 (cfg/common/ctc-common.cfg) */
  int32_t occupancy_precision;           /* atlas size / occupancy video size */
  int32_t map_count, absolute_d1, remove_duplicate_points, threshold_lossy_om;
  /* geometry smoothing of the decoder's post-processing (PCCDecoder.cpp:434-437 -> PCCCodec::smoothPointCloudPostprocess with gridSmoothing, PCCCodec.cpp:52-145, :980-1104;
   * the CTC switches it on: cfg/common/ctc-common.cfg:57-60 flagGeometrySmoothing 1, gridSmoothing 1, gridSize 8, thresholdSmoothing 64). 0 = off. */
  int32_t geometry_smoothing, grid_size, threshold_smoothing;
} oracle_atlas;
typedef struct { int n; int16_t* xyz; uint16_t* yuv; uint8_t* occupancy_map; uint32_t* block_to_patch; int n_smoothed; } oracle_cloud;   /* n_smoothed: points the geometry smoothing moved */
typedef struct { int n_a, n_b; uint64_t sse_ab, sse_ba, max_ab, max_ba; float mse_ab, mse_ba, psnr_ab, psnr_ba, psnr; } oracle_d1_result;

/* occ: occupancy video luma (width / precision x height / precision); d0, d1: luma of the two geometry maps (width x height samples of geo_bd
 * bits); t0, t1: attribute pictures, planar 4:2:0 (may be NULL) */
int oracle_reconstruct(const oracle_atlas* a, const oracle_patch* patches, int n_patches, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, int geo_bd,
                       const uint16_t* t0, const uint16_t* t1, int attr_bd, oracle_cloud* out);
void oracle_cloud_free(oracle_cloud* c);
int oracle_d1(const int16_t* a, int na, const int16_t* b, int nb, int peak, oracle_d1_result* out);
/* Point-to-plane (D2) metric, QualityMetrics::compute with computeC2p_ (PCCMetrics.cpp:100-124, :213-215, symmetric :299-309) between a source cloud A that comes
 * with normals and a decoded cloud B that gets its normals from A the way PCCMetrics::compute arranges it (PCCMetrics.cpp:371-376: copyNormals on the source,
 * scaleNormals on the reconstruction, PCCPointSet.cpp:2322-2380). Normals are fixed point, Q14 (16384 = 1.0), three per point of A.
 * Differences from the reference, all of them where it depends on the order its kd-tree returns equidistant points in: duplicates are merged first (as for D1) and
 * a merged point keeps the normal of its lowest-index duplicate (the reference copies normals by index into the deduplicated cloud); "the points at the same
 * distance" are ALL points at exactly the nearest squared distance (the reference looks at 5, 10, .. 30 results and stops there). */
typedef struct { int n_a, n_b; double sse_ab, sse_ba, max_ab, max_ba; float mse_ab, mse_ba, psnr_ab, psnr_ba, psnr; } oracle_d2_result;
int oracle_d2(const int16_t* a, const int16_t* normals_a, int na, const int16_t* b, int nb, int peak, oracle_d2_result* out);
#endif
