// chain.h -- launch interface of the layer-persistent decode-block experiment (chain.hip); a tools/debug build target, not
// part of the product library.
#pragma once
#include "../../../mlx_parallm_amd/csrc/kernels.h"

namespace mi {

// The linear layers of one decoder block's decode step (<= 8 rows, dense 16-bit weights) as ONE persistent launch with a
// weight-loader wave per CU (chain.hip): ops[i] reads what ops[i - 1] wrote when wait_prev[i] != 0.  `ctr`: CHAIN_CTR_WORDS
// zero-initialised words that live as long as launches use them; `base`: chain_grid() x the number of earlier launches on
// those counters (unsigned wrap-around is fine); *error is set non-zero if a bounded wait gave up.
constexpr int CHAIN_CTR_WORDS = 4 * 8 * 32;
bool chain_linear_ok(const LinearW& W, const GemvCall& c);
int chain_grid();
int chain_sq_ld(int K);          // row stride of the [8][ld] float table a residual linear leaves for the RMSNorm behind it (zeroed once)
int launch_chain(const LinearW* const* W, const GemvCall* calls, const int* wait_prev, int nops, int M, int act,
                 unsigned* ctr, unsigned base, unsigned spin_limit, int* error, hipStream_t st);

}  // namespace mi
