// Tap geometry of image_warp, shared by warp.hip and stacked.hip.
#pragma once
#include "common.h"

namespace unflow {

// image_warp: floor the FLOW, clamp the taps (core/image_warp.py:26-54).
struct ClampTaps {
  long long ia, ib, ic, id;  // pixel indices inside the item
  float xw, yw;
};

__device__ inline ClampTaps clamp_taps(int pos_x, int pos_y, float2 f, int H, int W) {
  ClampTaps t;
  const float flx = floorf(f.x), fly = floorf(f.y);
  t.xw = f.x - flx;
  t.yw = f.y - fly;
  int x0 = pos_x + (int)flx, y0 = pos_y + (int)fly;
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), W - 1);
  x1 = min(max(x1, 0), W - 1);
  y0 = min(max(y0, 0), H - 1);
  y1 = min(max(y1, 0), H - 1);
  t.ia = (long long)y0 * W + x0;
  t.ib = (long long)y1 * W + x0;
  t.ic = (long long)y0 * W + x1;
  t.id = (long long)y1 * W + x1;
  return t;
}

}  // namespace unflow
