// Anyres image preprocessing on the device: the integer arithmetic of Pillow's 8-bit LANCZOS
// resample (third-party code behind ref:vis_zephyr/model/multi_scale_process.py:88,160-163 `image.resize(..., LANCZOS)`;
// Pillow src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc) and the letterbox +
// 336 x 336 tiling + CLIP normalisation of ref :70-171.  Byte work, HBM / latency bound: one thread per output pixel
// (3 channels), taps read through L1/L2 (neighbouring outputs share most of their taps), 16-byte stores on the way out.
//
// The per-output-pixel weights are computed on the host exactly as Pillow does (double precision windowed sinc,
// normalised, rounded to int32 at 22 fractional bits: vz_hip/preprocess.py) - they depend on the two sizes only and are
// cached; the kernels below reproduce Pillow's accumulation: acc = 1 << 21; acc += pixel * k; out = clip8(acc >> 22).
#include "vz_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int acc) {
    const int v = acc >> PRECISION_BITS;            // arithmetic shift, as the C source
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y][xx][c] = clip8(sum_j src[y][lo + j][c] * k[xx][j]);  src [h][w][3], dst [h][w2][3]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, int h, int w, unsigned char* __restrict__ dst,
                                                         int w2, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
    const long total = (long)h * w2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % w2), y = (int)(i / w2);
        const int lo = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int* k = coefs + (size_t)xx * ksize;
        const unsigned char* p = src + ((size_t)y * w + lo) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int j = 0; j < n; ++j) {
            const int kj = k[j];
            a0 += (int)p[3 * j] * kj; a1 += (int)p[3 * j + 1] * kj; a2 += (int)p[3 * j + 2] * kj;
        }
        unsigned char* o = dst + (size_t)i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// dst[yy][x][c] = clip8(sum_j src[lo + j][x][c] * k[yy][j]);  src [h][w][3], dst [h2][w][3]
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ src, int h, int w, unsigned char* __restrict__ dst,
                                                         int h2, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
    const long total = (long)h2 * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w), yy = (int)(i / w);
        const int lo = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int* k = coefs + (size_t)yy * ksize;
        const unsigned char* p = src + ((size_t)lo * w + x) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int j = 0; j < n; ++j) {
            const int kj = k[j];
            const unsigned char* q = p + (size_t)j * w * 3;
            a0 += (int)q[0] * kj; a1 += (int)q[1] * kj; a2 += (int)q[2] * kj;
        }
        unsigned char* o = dst + (size_t)i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// out [1 + gw*gh][3][side][side] bf16: tile 0 = the global view, tile 1 + ty*gw + tx = crop (tx, ty) of the black canvas
// [gh*side][gw*side] that carries the resized image at (paste_x, paste_y).  lut [3][256] bf16 = CLIP rescale + normalise.
// One thread = 8 consecutive x of one (tile, channel, row): a 16-byte store.
__global__ __launch_bounds__(256) void anyres_tiles_kernel(const unsigned char* __restrict__ glob, const unsigned char* __restrict__ resized,
                                                           int nh, int nw, int paste_x, int paste_y, int gw, int gh, int side,
                                                           const unsigned short* __restrict__ lut, unsigned short* __restrict__ out) {
    __shared__ unsigned short sl[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) sl[i] = lut[i];
    __syncthreads();
    const int xchunks = side / 8;
    const long total = (long)(1 + gw * gh) * 3 * side * xchunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xc = (int)(i % xchunks);
        long r = i / xchunks;
        const int y = (int)(r % side); r /= side;
        const int c = (int)(r % 3);
        const int t = (int)(r / 3);
        u16x8 o;
        if (t == 0) {
            const unsigned char* p = glob + ((size_t)y * side + xc * 8) * 3 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = sl[c * 256 + p[3 * j]];
        } else {
            const int ty = (t - 1) / gw, tx = (t - 1) % gw;
            const int cy = ty * side + y - paste_y;           // row in the resized image
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cx = tx * side + xc * 8 + j - paste_x;
                const bool in = cy >= 0 && cy < nh && cx >= 0 && cx < nw;
                const int v = in ? resized[((size_t)cy * nw + cx) * 3 + c] : 0;      // letterbox bars are black (0, 0, 0)
                o[j] = sl[c * 256 + v];
            }
        }
        *(u16x8*)(out + (((size_t)t * 3 + c) * side + y) * side + xc * 8) = o;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
// ---- ViP "point" overlay (second half) ---------------------------------------------------------
// `image_blending(shape="point")` (ref:vis_zephyr/model/vip_processor/conversation_generator.py:143-153,170-175 ->
// ref:vis_zephyr/model/vip_processor/shape_draw.py:130-138) = Pillow's filled ellipse on a transparent RGBA canvas +
// Image.alpha_composite + convert("RGB").  Both are integer arithmetic (Pillow src/libImaging/Draw.c ellipseNew / quarter_next,
// src/libImaging/AlphaComposite.c), restated in oracle/vip_oracle.py and pinned against Pillow; reproduced here bit for bit.
// One launch per point, over the ellipse's bounding box: lane 0 of every workgroup walks the quarter ellipse in doubled
// coordinates (<= (a + b) / 2 + 1 error-minimising steps, int64) into an LDS table r[(Y - b % 2) / 2] = half-width of scanline
// Y; then a pixel (x, y) is inside iff (a - r) >> 1 <= x - x0 <= (a + r) >> 1 with Y = |2 (y - y0) - b|, and is replaced by
// the composite of the constant colour over it: t = src * (sa * 128) + dst * ((255 - sa) * 128) + (0x80 << 7);
// out = ((((t >> 8) + t) >> 8) >> 7).
constexpr int VIP_MAX_ROWS = 2048;        // doubled-scanline table: ellipses up to 4094 pixels tall

__global__ __launch_bounds__(256) void vip_point_kernel(unsigned char* __restrict__ img, int h, int w, int x0, int y0, int a, int b,
                                                        unsigned rgba) {
    __shared__ int r_of[VIP_MAX_ROWS];
    if (threadIdx.x == 0) {
        const long long a2 = (long long)a * a, b2 = (long long)b * b, a2b2 = a2 * b2;
        auto delta = [&](long long x, long long y) { const long long d = a2 * y * y + b2 * x * x - a2b2; return d < 0 ? -d : d; };
        int cx = a, cy = b & 1;
        const int ex = a & 1, ey = b;
        int last_row = -1;
        while (true) {
            const int row = (cy - (b & 1)) >> 1;
            if (row != last_row) { r_of[row] = cx; last_row = row; }       // the FIRST point of a scanline is its right end
            if (cx == ex && cy == ey) break;
            int nx = cx, ny = cy + 2;
            long long nd = delta(nx, ny);
            if (nx > 1) {
                long long d = delta(cx - 2, cy + 2);
                if (nd > d) { nx = cx - 2; ny = cy + 2; nd = d; }
                d = delta(cx - 2, cy);
                if (nd > d) { nx = cx - 2; ny = cy; }
            }
            cx = nx; cy = ny;
        }
    }
    __syncthreads();
    const unsigned sr = rgba & 255, sg = (rgba >> 8) & 255, sb = (rgba >> 16) & 255, sa = rgba >> 24;
    const unsigned c1 = sa * 128u, c2 = (255u - sa) * 128u;
    const int bw = (a >> 1) + 1 + ((a & 1) ? 1 : 0), bh = b + 1;         // pixel box [x0, x0 + a] x [y0, y0 + b] (spans end at (a + r) >> 1 <= a)
    const int cols = a + 1;
    (void)bw;
    const long total = (long)cols * bh;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / cols), dx = (int)(i - (long)t * cols);
        const int y = y0 + t, x = x0 + dx;
        if (y < 0 || y >= h || x < 0 || x >= w) continue;
        int Y = 2 * t - b;
        Y = Y < 0 ? -Y : Y;
        const int r = r_of[(Y - (b & 1)) >> 1];
        if (dx < ((a - r) >> 1) || dx > ((a + r) >> 1)) continue;
        unsigned char* p = img + ((size_t)y * w + x) * 3;
        unsigned tr = sr * c1 + p[0] * c2 + (0x80u << 7), tg = sg * c1 + p[1] * c2 + (0x80u << 7), tb = sb * c1 + p[2] * c2 + (0x80u << 7);
        p[0] = (unsigned char)((((tr >> 8) + tr) >> 8) >> 7);
        p[1] = (unsigned char)((((tg >> 8) + tg) >> 8) >> 7);
        p[2] = (unsigned char)((((tb >> 8) + tb) >> 8) >> 7);
    }
}

}  // namespace


extern "C" int vz_op_resample_u8(const void* d_src, int h, int w, void* d_tmp, void* d_dst, int h2, int w2, const int* d_xbounds,
                                 const int* d_xcoefs, int kx, const int* d_ybounds, const int* d_ycoefs, int ky, vz_stream stream) {
    VZ_CHECK_ARG(d_src && d_dst && h > 0 && w > 0 && h2 > 0 && w2 > 0, "resample: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const bool need_h = w2 != w, need_v = h2 != h;
    VZ_CHECK_ARG(!need_h || (d_xbounds && d_xcoefs && kx > 0), "resample: horizontal pass needs its bounds / coefficients");
    VZ_CHECK_ARG(!need_v || (d_ybounds && d_ycoefs && ky > 0), "resample: vertical pass needs its bounds / coefficients");
    VZ_CHECK_ARG(!(need_h && need_v) || d_tmp, "resample: two passes need the [h, w2, 3] intermediate buffer");
    if (!need_h && !need_v) {       // Pillow returns a copy
        VZ_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)h * w * 3, hipMemcpyDeviceToDevice, s));
        return VZ_OK;
    }
    const unsigned char* cur = (const unsigned char*)d_src;
    if (need_h) {                   // Pillow: horizontal pass first, 8-bit intermediate
        unsigned char* o = (unsigned char*)(need_v ? d_tmp : d_dst);
        hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((long)h * w2)), dim3(256), 0, s, cur, h, w, o, w2, d_xbounds, d_xcoefs, kx);
        VZ_LAUNCH_CHECK();
        cur = o;
    }
    if (need_v) {
        hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((long)h2 * w2)), dim3(256), 0, s, cur, h, w2, (unsigned char*)d_dst, h2,
                           d_ybounds, d_ycoefs, ky);
        VZ_LAUNCH_CHECK();
    }
    return VZ_OK;
}

extern "C" int vz_op_anyres_tiles(const void* d_global, const void* d_resized, int nh, int nw, int paste_x, int paste_y, int grid_w,
                                  int grid_h, int side, const void* d_lut, void* d_out, vz_stream stream) {
    VZ_CHECK_ARG(d_global && d_resized && d_lut && d_out && nh > 0 && nw > 0 && grid_w > 0 && grid_h > 0 && side > 0 && side % 8 == 0,
                 "anyres_tiles: bad argument");
    VZ_CHECK_ARG(paste_x >= 0 && paste_y >= 0 && paste_x + nw <= grid_w * side && paste_y + nh <= grid_h * side,
                 "anyres_tiles: the resized image does not fit the %d x %d canvas", grid_w * side, grid_h * side);
    VZ_CHECK_ARG(((uintptr_t)d_out & 15) == 0, "anyres_tiles: output must be 16-byte aligned");
    const long total = (long)(1 + grid_w * grid_h) * 3 * side * (side / 8);
    hipLaunchKernelGGL(anyres_tiles_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)d_global,
                       (const unsigned char*)d_resized, nh, nw, paste_x, paste_y, grid_w, grid_h, side, (const unsigned short*)d_lut,
                       (unsigned short*)d_out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

extern "C" int vz_op_vip_point(void* d_image, int h, int w, int x0, int y0, int x1, int y1, unsigned rgba, vz_stream stream) {
    VZ_CHECK_ARG(d_image && h > 0 && w > 0, "vip_point: bad argument");
    const int a = x1 - x0, b = y1 - y0;
    if (a < 0 || b < 0 || (rgba >> 24) == 0) return VZ_OK;          // Pillow draws nothing / alpha 0 composites to the image itself
    VZ_CHECK_ARG(b / 2 + 1 <= VIP_MAX_ROWS && a <= 1 << 20, "vip_point: ellipse %d x %d too large", a, b);
    const long total = (long)(a + 1) * (b + 1);
    hipLaunchKernelGGL(vip_point_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)d_image, h, w, x0, y0, a, b, rgba);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
