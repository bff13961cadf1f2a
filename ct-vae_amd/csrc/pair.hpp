// Paired backward launch (ctvae_conv_backward): while a PairCtx is installed for the calling thread, the data-gradient
// launcher (tapgemm_fast.hip, 64x64 pipelined tile kernel) and the weight-gradient launcher (wgrad.hip, lean 64x64 kernel)
// RECORD their main launch instead of issuing it, and every finishing launch behind them (split-K finish, slab
// reduction) is queued; pair_flush() then issues ONE conv_bwd_pair_kernel for both GEMMs (or the single recorded kernel,
// if only one side took its pairable path) followed by the queued launches in order.  Launches of any other path are
// issued immediately as usual -- the two GEMMs are independent and work in disjoint halves of the workspace.
#pragma once
#include <functional>
#include <vector>

#include "finish.hpp"
#include "tapgemm.hpp"
#include "wgrad_fast.hpp"

namespace ctvae {

struct PairCtx {
  bool haveA = false, haveB = false;
  long dgrad_wgs = 0;          // workgroups of the data gradient and K chunks of its longest one (planned before the weight
  int dgrad_chunks = 0;        // gradient is sized)
  TapGemmArgs A;
  unsigned gxA = 0, gyA = 0, gzA = 0;
  double flopsA = 0, bytesA = 0;
  WgradArgs B;
  int lgQw = -1, lgQhw = -1, lgC = -1;
  unsigned gxB = 0, gyB = 0;
  double flopsB = 0, bytesB = 0;
  // finishing passes: the data gradient's split-K sum and the weight gradient's slab reduction share a launch when both exist
  bool haveSK = false, haveRed = false;
  SplitKJob sk;
  ReduceJob j1, j2;
  int nb1 = 0, nb2 = 0, accumulate = 0;
  double bytesSK = 0, bytesRed = 0;
  // BatchNorm-backward finalize of the layer below (its sums come out of the data gradient's epilogues): rides in the
  // finishing launch when there is one, else pair_flush() issues it on its own
  bool haveBF = false;
  BnFinJob bf;
  std::vector<std::function<int()>> later;
};

PairCtx*& pair_ctx();                                    // tapgemm_fast.hip (thread-local, null = not pairing)
int pair_flush(PairCtx& c, hipStream_t st);              // tapgemm_fast.hip
int launch_wgrad_fast_recorded(const PairCtx& c, hipStream_t st);   // wgrad.hip: the recorded weight-gradient kernel on its own
int launch_finish_recorded(const PairCtx& c, hipStream_t st);       // wgrad.hip: recorded reduction (+ split-K finish) in one launch
int launch_splitk_recorded(const PairCtx& c, hipStream_t st);       // tapgemm.hip: recorded split-K finish on its own
int launch_bn_bwd_finalize_job(const BnFinJob& j, hipStream_t st);  // bn.hip: the finalize job as a launch of its own

}  // namespace ctvae
