// Translation unit of the layer-at-a-time family (any latent_dim / mlp_dim: the 1M ... 317M world models, c3 / c4 / c5):
// the GEMM and row kernels (layered_kernels.cuh: exact fp32; layered_split.cuh: f16x2 split) and their host orchestration
// (layered_host.cuh), behind the lay_* functions of launch.h.
#include "launch.h"

namespace {
#include "fused_kernels.cuh"  // device helpers shared with the fused family: ldw, SPLIT_MFMA, mish_fast, split4
#include "tile_order.h"
#include "layered_kernels.cuh"
#include "layered_split.cuh"
#include "layered_wide.cuh"
#include "layered_mid.cuh"
}  // namespace

namespace tdk {
#include "layered_host.cuh"
}  // namespace tdk
