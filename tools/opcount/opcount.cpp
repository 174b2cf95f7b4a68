// FLOP counter build of the oracle: the oracle's C source compiled as C++ with `real` = CReal (counted.h).
#include "counted.h"
thread_local OpCounts g_ops = {0, 0, 0, 0, 0, 0};
#define DIAL_OPCOUNT 1
#define REAL CReal
extern "C" {
#include "../../oracle/dial_oracle.c"
void opcount_reset(void) { g_ops = OpCounts{0, 0, 0, 0, 0, 0}; }
void opcount_get(unsigned long long* out) {
  out[0] = g_ops.add; out[1] = g_ops.mul; out[2] = g_ops.div; out[3] = g_ops.sqrt_; out[4] = g_ops.trans; out[5] = g_ops.cmp;
}
}
