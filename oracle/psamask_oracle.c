/* TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of the reference's PSA mask operator
 *   /root/reference/lib/psa/src/cpu/psamask.cpp:11-113  (the four loop nests)
 *   /root/reference/lib/psa/src/cpu/psamask.cpp:115-133 (psa_type dispatch: 0 collect, else distribute)
 * written from the index maps, not copied: one generic walker parameterised by direction and type.
 * Pinned bit-exactly against the compiled reference (oracle/_ref, see oracle/build_ref.py) and the
 * fixtures in tests/golden/psamask_*.npz by tests/test_oracle.py.
 *
 * NCHW fp32.  mask tensor  M[n][hi*mW+wi][h][w]      (num, mH*mW, H, W)
 *             buffer tensor B[n][p][q]                (num, H*W,  H*W)
 *   collect   : p = (h+hi-hh)*W + (w+wi-hw), q = h*W + w        (psamask.cpp:26-29)
 *   distribute: p = h*W + w,                 q = (h+hi-hh)*W + (w+wi-hw)   (psamask.cpp:52-55)
 * forward copies M -> B on the in-window taps, backward copies B -> M (psamask.cpp:78-79,104-105).
 * Destination must be zero-filled by the caller (lib/psa/functions/psamask.py:17,31).
 */
#include <stddef.h>

static void walk(int backward, int psa_type, int num, int H, int W, int mH, int mW, int hh, int hw,
                 const float* src, float* dst) {
  const size_t HW = (size_t)H * W;
  for (int n = 0; n < num; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        /* window of mask taps whose shifted position stays inside the feature map (psamask.cpp:20-23) */
        const int hs = hh - h > 0 ? hh - h : 0;
        const int he = mH < H + hh - h ? mH : H + hh - h;
        const int ws = hw - w > 0 ? hw - w : 0;
        const int we = mW < W + hw - w ? mW : W + hw - w;
        for (int hi = hs; hi < he; ++hi)
          for (int wi = ws; wi < we; ++wi) {
            const size_t self = (size_t)h * W + w;
            const size_t shifted = (size_t)(h + hi - hh) * W + (w + wi - hw);
            const size_t p = psa_type == 0 ? shifted : self;
            const size_t q = psa_type == 0 ? self : shifted;
            const size_t bi = ((size_t)n * HW + p) * HW + q;
            const size_t mi = (((size_t)n * mH * mW + (size_t)hi * mW + wi) * H + h) * W + w;
            if (backward) dst[mi] = src[bi];
            else dst[bi] = src[mi];
          }
      }
}

void oracle_psamask_forward(int psa_type, const float* input, float* output, int num, int H, int W,
                            int mH, int mW, int hh, int hw) {
  walk(0, psa_type, num, H, W, mH, mW, hh, hw, input, output);
}

void oracle_psamask_backward(int psa_type, const float* grad_output, float* grad_input, int num, int H,
                             int W, int mH, int mW, int hh, int hw) {
  walk(1, psa_type, num, H, W, mH, mW, hh, hw, grad_output, grad_input);
}
