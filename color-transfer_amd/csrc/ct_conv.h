// ct_conv.h -- argument block of the LDS-tiled MFMA convolution (cnn.hip), shared with gmflow.hip
#pragma once
#include <hip/hip_runtime.h>

namespace ct {

struct ConvArgs {
    const float *in;
    const float *in2;       // split kernel only: channels >= cin1 come from this tensor (torch.cat on dim 1 without the copy); or NULL
    int cin1;               // channels of `in` when in2 != NULL (a multiple of 16)
    long long in2_bstride;
    const float *wp;
    const float *bias;      // [MT*32], zero padded
    const float *residual;  // nullable, same shape/strides as out
    float *out;
    int cin, cout, H, W;
    long long in_bstride, out_bstride, res_bstride;   // elements between images of a batch
    int act;     // 0 none, 1 LeakyReLU(0.01), 2 ReLU, 3 sigmoid, 4 tanh
    int clamp;   // 1 = clamp to [0,1]
    int groups;  // output-channel groups of MT*32 (weights packed group-major); 1 for cout <= 64
    unsigned long long *prof;   // diagnostic builds only (CT_CONV_PROFILE); NULL otherwise
    int n_images = 0;           // conv_ws only (set by its launcher)
    const float *in3 = nullptr;  // split kernel only: channels >= cin2 come from this tensor (a third torch.cat operand); cin2 % 16 == 0
    int cin2 = 0;
    long long in3_bstride = 0;
    int f16 = 0;                // conv_ws only: 1 = two fp16 pieces (weights scaled by 2^w_exp), 0 = three bf16 pieces
    int w_exp = 0;
    int rows_channels = 0;      // split kernel only: > 0 = `out` is a token-rows tensor [N*H, W, rows_channels] and the result goes to its
    int post_op = 0;            // split kernel, F16 form, NCHW output: 1 = result * p1; 2 = (1 - p1) * p2 + p1 * result (the GRU gate of
    const float *p1 = nullptr;  // reg_refine.py:46-55: p1 = z, p2 = h), applied after the activation; p1 / p2 laid out like `out`
    const float *p2 = nullptr;
    int res_pre = 0;            // split kernel only: 1 = `residual` is added BEFORE the activation (a pre-computed partial convolution)
    int rows_c0 = 0;            // channels rows_c0 .. rows_c0 + cout (multiples of 4); no residual / clamp in this mode
    float *sk_ws = nullptr;         // split kernel, F16 form: stream-K scratch (partial tiles, one 64 KiB slot per workgroup), or NULL
    unsigned int *sk_flags = nullptr;   // ... and its flags (one per workgroup; zero between launches)
    unsigned int sk_epoch = 1;      // ... the value a producer of THIS launch publishes (non-zero, new for every launch: a flag left behind
                                    //     by a producer that came too late for its consumer cannot satisfy the next launch's wait)
};

struct GConvArgs {
    const float *in, *wp, *bias;   // wp: [coutp/64][kh*kw][cin_pairs][2][64], coutp = 64*ceil(cout/64); bias padded to coutp (or null)
    float *out;
    int cin, cout, coutp, H, W, Ho, Wo, KH, KW, stride, padH, padW;
    long long in_bstride, out_bstride;
    int act;      // 0 none, 1 LeakyReLU(0.01), 2 ReLU, 3 sigmoid, 4 tanh, 5 swish
    int cchunk;   // input channels staged per LDS pass (even)
};

// conv_direct.hip: float32 VALU convolutions for the shapes an implicit-GEMM tile wastes (returns 1 if the shape is not theirs):
//   cout <= 4, kernel 3x3 / 1x1, stride 1, "same" padding (GMFlow's flow head: 256 -> 2), and
//   cin <= 3, kernel <= 7x7, stride 1 or 2 (its 7x7 stem 3 -> 64 and the motion encoder's 7x7 on the 2-channel flow)
int conv_direct(const GConvArgs &a, int N, hipStream_t s);

// cnn.hip: stride-1, padding k/2, kernel 3x3 / 1x1 / 1x5 / 5x1, weights packed [group][tap][cin_pair][2][64];
// returns CT_OK / an error, or 1 if (kh, kw) has no LDS-tiled kernel
int conv_fast(const ConvArgs &a, int N, int kh, int kw, hipStream_t s);

// conv_ws.hip: 3x3, 32 < cin <= 64, one input tensor: weights stationary in registers, contraction split across the waves;
// returns CT_OK / an error, or 1 if the geometry is not this kernel's
int conv_ws(const ConvArgs &a, int N, bool gen, hipStream_t s);

// conv_wino.hip / conv_wino4.hip: the same convolution as Winograd F(2x2, 3x3) on two fp16 pieces (weights packed by
// ct_hip.pack_conv_weight_wino16); 1 = not this kernel's geometry
int conv_wino(const ConvArgs &a, int N, hipStream_t s);
int conv_wino4(const ConvArgs &a, int N, hipStream_t s);

}  // namespace ct
