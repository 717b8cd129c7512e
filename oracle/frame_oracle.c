/*
 * frame_oracle.c — CPU restatement of the reference's frame pipeline between the emulator and
 * the network input.  TEST INFRASTRUCTURE ONLY.
 *
 *   MaxAndSkipEnv.step   parl/env/atari_wrappers.py:223-240   max_frame = obs_buffer.max(axis=0)
 *   WarpFrame.observation parl/env/atari_wrappers.py:263-267  cv2.cvtColor(RGB2GRAY) then
 *                                                              cv2.resize((dim,dim), INTER_AREA)
 *
 * cv2 is third-party and absent here ("parity unpinned"), so the two OpenCV
 * routines are restated from OpenCV's published algorithm (imgproc/color_rgb.cpp RGB2Gray<uchar>
 * and imgproc/resize.cpp computeResizeAreaTab / ResizeArea_Invoker, 8UC1, non-integer scale):
 *   gray = (R*4899 + G*9617 + B*1868 + (1<<13)) >> 14
 *     — the 14-bit coefficients of OpenCV <= 4.3 (yuv_shift = 14: R2Y = 4899, G2Y = 9617, B2Y = 1868), the
 *     version the reference's CI pins (.teamcity/requirements.txt:3: opencv-python<=4.3.0.34).  OpenCV >= 4.4
 *     uses 15 bits (9798 / 19235 / 3735, >> 15).  The two agree on every one of the 128 NTSC palette colours,
 *     i.e. on every single frame; they differ by 1 on 12 of the 16,384 per-channel maxima of two palette
 *     colours that MaxAndSkipEnv can produce (tests/test_frame_oracle_pin.py enumerates them).
 *   area resize: per-axis (src index, alpha) tap tables built in double and stored as float;
 *   for each source row: buf[dx] = sum_k S[sx_k]*alpha_k (float, in tap order);
 *   sum[dx] = beta_0*buf_0 (assignment) then += beta_j*buf_j; dst = saturate(cvRound(sum)).
 * All float ops are separate multiplies and adds (x86-64 OpenCV builds do not contract to FMA):
 * compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern const uint32_t atari_ntsc_palette[128];

#define SRC_W 160
#define SRC_H 210

/* table blob layout (shared with the device, see include/parl_hip.h):
 * int32 hdr[8] = {dim, nx, ny, off_xstart, off_ystart, off_xtap, off_ytap, total_bytes}
 * int32 xstart[dim+1], ystart[dim+1]; taps: {int32 si; float alpha;}[nx], [ny]; uint32 pal[128] */
typedef struct { int32_t si; float alpha; } Tap;

static int area_tab(int ssize, int dsize, int32_t* start, Tap* tab) {
  const double inv = (double)dsize / (double)ssize;
  const double scale = 1.0 / inv;
  int k = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    start[dx] = k;
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cell = scale < (ssize - fsx1) ? scale : (ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    if (sx1 - fsx1 > 1e-3) {
      if (tab) { tab[k].si = sx1 - 1; tab[k].alpha = (float)((sx1 - fsx1) / cell); }
      k++;
    }
    for (int sx = sx1; sx < sx2; ++sx) {
      if (tab) { tab[k].si = sx; tab[k].alpha = (float)(1.0 / cell); }
      k++;
    }
    if (fsx2 - sx2 > 1e-3) {
      double r = fsx2 - sx2;
      if (r > 1.0) r = 1.0;
      if (r > cell) r = cell;
      if (tab) { tab[k].si = sx2; tab[k].alpha = (float)(r / cell); }
      k++;
    }
  }
  start[dsize] = k;
  return k;
}

size_t oracle_frame_tables_bytes(int dim) {
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(dim + 1));
  int nx = area_tab(SRC_W, dim, tmp, 0);
  int ny = area_tab(SRC_H, dim, tmp, 0);
  free(tmp);
  return 8 * 4 + 2 * (size_t)(dim + 1) * 4 + (size_t)(nx + ny) * sizeof(Tap) + 128 * 4;
}

int oracle_frame_tables_init(void* blob, int dim) {
  int32_t* hdr = (int32_t*)blob;
  int32_t* xstart = hdr + 8;
  int32_t* ystart = xstart + dim + 1;
  Tap* xt = (Tap*)(ystart + dim + 1);
  int nx = area_tab(SRC_W, dim, xstart, xt);
  Tap* yt = xt + nx;
  int ny = area_tab(SRC_H, dim, ystart, yt);
  uint32_t* pal = (uint32_t*)(yt + ny);
  memcpy(pal, atari_ntsc_palette, 128 * 4);
  hdr[0] = dim; hdr[1] = nx; hdr[2] = ny;
  hdr[3] = (int32_t)((char*)xstart - (char*)blob);
  hdr[4] = (int32_t)((char*)ystart - (char*)blob);
  hdr[5] = (int32_t)((char*)xt - (char*)blob);
  hdr[6] = (int32_t)((char*)yt - (char*)blob);
  hdr[7] = (int32_t)((char*)(pal + 128) - (char*)blob);
  return 0;
}

static inline uint8_t gray_of(uint32_t r, uint32_t g, uint32_t b) {
  return (uint8_t)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
}

static inline uint8_t sat_round(float v) {
  /* cv::saturate_cast<uchar>(float) = cvRound (round half to even) then clamp */
  long r = lrintf(v);
  return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

/* gray[210*160] from one or two frames.  fmt 0: RGB u8 [210,160,3]; fmt 1: TIA colour bytes. */
static void gray_max(const uint8_t* f0, const uint8_t* f1, int fmt, uint8_t* gray) {
  for (int i = 0; i < SRC_W * SRC_H; ++i) {
    uint32_t r, g, b;
    if (fmt == 0) {
      r = f0[3 * i]; g = f0[3 * i + 1]; b = f0[3 * i + 2];
      if (f1) {
        if (f1[3 * i] > r) r = f1[3 * i];
        if (f1[3 * i + 1] > g) g = f1[3 * i + 1];
        if (f1[3 * i + 2] > b) b = f1[3 * i + 2];
      }
    } else {
      uint32_t c = atari_ntsc_palette[f0[i] >> 1];
      r = (c >> 16) & 255; g = (c >> 8) & 255; b = c & 255;
      if (f1) {
        uint32_t d = atari_ntsc_palette[f1[i] >> 1];
        uint32_t r1 = (d >> 16) & 255, g1 = (d >> 8) & 255, b1 = d & 255;
        if (r1 > r) r = r1;
        if (g1 > g) g = g1;
        if (b1 > b) b = b1;
      }
    }
    gray[i] = gray_of(r, g, b);
  }
}

/* frames0/frames1: [E, ...]; out: E frames of dim*dim bytes, out_stride bytes apart. */
int oracle_frame_post_u8(const uint8_t* frames0, const uint8_t* frames1, int fmt, uint8_t* out,
                         int64_t out_stride, int E, int dim, const void* blob) {
  const int32_t* hdr = (const int32_t*)blob;
  if (hdr[0] != dim) return -1;
  const int32_t* xstart = (const int32_t*)((const char*)blob + hdr[3]);
  const int32_t* ystart = (const int32_t*)((const char*)blob + hdr[4]);
  const Tap* xt = (const Tap*)((const char*)blob + hdr[5]);
  const Tap* yt = (const Tap*)((const char*)blob + hdr[6]);
  const size_t fsz = (size_t)SRC_W * SRC_H * (fmt == 0 ? 3 : 1);
  uint8_t* gray = (uint8_t*)malloc(SRC_W * SRC_H);
  for (int e = 0; e < E; ++e) {
    gray_max(frames0 + e * fsz, frames1 ? frames1 + e * fsz : 0, fmt, gray);
    uint8_t* o = out + (size_t)e * (size_t)out_stride;
    for (int dy = 0; dy < dim; ++dy) {
      for (int dx = 0; dx < dim; ++dx) {
        float sum = 0.0f;
        for (int j = ystart[dy]; j < ystart[dy + 1]; ++j) {
          const uint8_t* S = gray + yt[j].si * SRC_W;
          float buf = 0.0f;
          for (int k = xstart[dx]; k < xstart[dx + 1]; ++k) buf += (float)S[xt[k].si] * xt[k].alpha;
          if (j == ystart[dy]) sum = yt[j].alpha * buf;
          else sum += yt[j].alpha * buf;
        }
        o[dy * dim + dx] = sat_round(sum);
      }
    }
  }
  free(gray);
  return 0;
}
