// Process-wide tuning options of libdq_hip.so (dq_set_option / dq_get_option, include/dq_hip.h).  Read at launch time; a change bumps
// options_epoch(), which the cached sampling graph of a plan records (a captured step has the dispatch of its capture time baked in).
#pragma once
#include <cstdint>

namespace dq {

enum Option {
  // rows from which Residual(PreNorm(LinearAttention)) over rows of 2 / 4 (/ 8) positions runs in the one-register-group-per-position form
  // (k_la_small.hip) instead of the register-resident form (k_linattn.hip).  < 0 (default): the device rule -- one 32-row tile per SIMD
  // (4 x compute units x 32 rows: 32,768 on MI355X); below that the launch is a latency chain and the shorter prologue wins
  OPT_LA_SMALL_MIN_ROWS = 0,
  // the same choice for the backward (k_la_rows_bwd.hip against k_la_bwd.hip); < 0 (default): every row count
  OPT_LA_ROWS_BWD_MIN_ROWS = 1,
  OPT_COUNT
};
int64_t option(Option o);
unsigned options_epoch();
int option_index(const char* key);  // -1: unknown
void set_option(int index, int64_t value);

}  // namespace dq
