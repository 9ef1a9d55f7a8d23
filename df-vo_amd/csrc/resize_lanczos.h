// Pillow-exact 8-bit LANCZOS resize on the device (the depth net's input stage).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dfvo {

// uint8 [H, W, 3] -> uint8 [oh, ow, 3]: horizontal pass into `tmp`, vertical pass into the destination, each with
// Pillow's 22-bit fixed-point coefficient tables (built on the host at init, resident on the device)
struct LanczosResizer {
    int H = 0, W = 0, oh = 0, ow = 0;
    int ksx = 0, ksy = 0;                 // coefficients per output column / row
    int *bx = nullptr, *kx = nullptr;     // [ow][2] (first input column, count), [ow][ksx]
    int *by = nullptr, *ky = nullptr;     // [oh][2], [oh][ksy]
    uint8_t* tmp = nullptr;               // [H][ow][3]
    int init(int H, int W, int oh, int ow);
    int enqueue(const uint8_t* d_src, uint8_t* d_dst, hipStream_t s) const;
    void release();
};

// cv2.resize(img, (ow, oh)) for uint8 [H, W, C] (INTER_LINEAR, OpenCV 3.4.3's 11-bit fixed point), utils.py:51
int enqueue_resize_linear_u8(const uint8_t* d_src, int H, int W, int C, uint8_t* d_dst, int oh, int ow, hipStream_t s, int pitch = 0,
                             int rev = 0);  // pitch: pixels per source row (0 = W); rev: reverse the channel order (BGR -> RGB)

int lanczos_coeffs_host(int in_size, int out_size, int* bounds, int* coeffs, int coeff_cap, int* ksize);

}  // namespace dfvo
