// Verification stage: reconstruction of the point cloud from decoded maps and the D1 metric. See rbt_pcc.cpp.
#pragma once
#include <string>
#include "../../include/rbt.h"
namespace rbt {
int pcc_reconstruct(std::string& err, const rbt_atlas_params* a, const rbt_patch* patches, int n_patches, const uint16_t* occ, const uint16_t* d0, const uint16_t* d1, int geo_bd,
                    const uint16_t* t0, const uint16_t* t1, int attr_bd, rbt_cloud* out);
int pcc_d1(std::string& err, const int16_t* a, int na, const int16_t* b, int nb, int peak, rbt_d1_result* out);
int pcc_d2(std::string& err, const int16_t* a, const int16_t* normals_a, int na, const int16_t* b, int nb, int peak, rbt_d2_result* out);
}
