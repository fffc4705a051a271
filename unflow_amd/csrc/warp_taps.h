// Tap geometry of image_warp, shared by warp.hip and stacked.hip.
#pragma once
#include "common.h"

namespace unflow {

// 2-D tiling of a 1-D grid-stride index: v enumerates the pixels of 64x4 tiles, tile by
// tile.  A 256-thread block then reads 5 image rows for its 4 output rows of bilinear
// taps instead of 2 rows for every output row, so neighbouring tap rows are shared
// inside the block's L1/L2 traffic instead of being fetched by another CU (often on
// another XCD, with its own L2) a second time.
constexpr int kTileW = 64, kTileH = 4;
struct TiledPixel {
  long long idx, b;   // linear pixel index b*H*W + y*W + x, batch item
  int x, y;
  bool ok;            // inside the image
};
__device__ inline TiledPixel tiled_pixel(long long v, int H, int W) {
  const int ntx = (W + kTileW - 1) / kTileW, nty = (H + kTileH - 1) / kTileH;
  const int lane = (int)(v & (kTileW * kTileH - 1));
  long long tile = v >> 8;
  static_assert(kTileW * kTileH == 256, "tile = 256 pixels");
  TiledPixel p;
  const int tx = (int)(tile % ntx);
  tile /= ntx;
  const int ty = (int)(tile % nty);
  p.b = tile / nty;
  p.x = tx * kTileW + (lane & (kTileW - 1));
  p.y = ty * kTileH + lane / kTileW;
  p.ok = p.x < W && p.y < H;
  p.idx = (p.b * H + p.y) * W + p.x;
  return p;
}
inline long long tiled_total(int B, int H, int W) {
  return (long long)B * ((H + kTileH - 1) / kTileH) * ((W + kTileW - 1) / kTileW) * (kTileW * kTileH);
}

// image_warp: floor the FLOW, clamp the taps (core/image_warp.py:26-54).
struct ClampTaps {
  long long ia, ib, ic, id;  // pixel indices inside the item
  float xw, yw;
};

// The clamped tap coordinates themselves (a scatter kernel wants (x, y), not the flat index).
struct ClampCoords {
  int x0, y0, x1, y1;
  float xw, yw;
};

__device__ inline ClampCoords clamp_tap_coords(int pos_x, int pos_y, float2 f, int H, int W) {
  ClampCoords t;
  const float flx = floorf(f.x), fly = floorf(f.y);
  t.xw = f.x - flx;
  t.yw = f.y - fly;
  const int x0 = pos_x + (int)flx, y0 = pos_y + (int)fly;
  t.x0 = min(max(x0, 0), W - 1);
  t.x1 = min(max(x0 + 1, 0), W - 1);
  t.y0 = min(max(y0, 0), H - 1);
  t.y1 = min(max(y0 + 1, 0), H - 1);
  return t;
}

__device__ inline ClampTaps clamp_taps(int pos_x, int pos_y, float2 f, int H, int W) {
  const ClampCoords c = clamp_tap_coords(pos_x, pos_y, f, H, W);
  ClampTaps t;
  t.xw = c.xw;
  t.yw = c.yw;
  t.ia = (long long)c.y0 * W + c.x0;
  t.ib = (long long)c.y1 * W + c.x0;
  t.ic = (long long)c.y0 * W + c.x1;
  t.id = (long long)c.y1 * W + c.x1;
  return t;
}

}  // namespace unflow
