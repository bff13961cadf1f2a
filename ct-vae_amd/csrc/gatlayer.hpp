// Argument blocks of the fused GATv2 layer kernels (gatlayer.hip); see include/ctvae_hip.h ctvae_gat_layer_*.
#pragma once
#include <hip/hip_runtime.h>

namespace ctvae {

struct GatLayerArgs {
  const float* xl;        // [B*64][ld], slot hs at columns hs*C .. hs*C+C-1
  const float* xr;        // same row stride
  int ld;
  const float* adj;       // [B][64][64] weighted adjacency, 0 = no edge
  const float* we;        // [H][C]  lin_edge.weight
  const float* att;       // [H][C]
  const float* bias;      // [H][C]
  const int* head_map;    // [B*Hs] or null
  float* out;             // [B*64][ldo], slot hs at columns hs*C ..
  int ldo;
  float* alpha;           // [B][Hs][64][64]
  int B, Hs, C;
  float slope;            // GATv2 negative_slope (0.2)
  int act;                // activation behind the layer: ACT_NONE or ACT_LRELU (nn.LeakyReLU between the two layers)
};

struct GatBwdArgs {
  GatLayerArgs f;
  const float* g_out;     // [B*64][ldo] gradient w.r.t. out
  float* dS;              // [B][Hs][64][64]
  float* dattr;           // [B][Hs][64][64]
  float* dxl;             // [B*64][ldd] slot hs at hs*C: receives the aggregation's share (gat_proj_bwd adds the rest)
  float* dxr;
  int ldd;
  float* dbias_part;      // [B][Hs][C]
  float* datt_part;       // [B][Hs][C]
  float* dwe_part;        // [B][Hs][C]
};

int launch_gat_layer_forward(const GatLayerArgs& a, hipStream_t st);
int launch_gat_layer_backward(const GatBwdArgs& p, float* dadj, int accumulate_dadj, hipStream_t st);

}  // namespace ctvae
