// Baseline JPEG decoding on the device (gfx950, MI355X): the decode half of the ingest row.
//
// Replaces pil_loader = Image.open(f).convert('RGB') (mdir/external/cirtorch/datasets/datahelpers.py:39-47, called per image from
// genericdataset.py:66-102).  The reference decodes through Pillow, i.e. libjpeg-turbo with its defaults; the result here is
// bit-identical to that: the "islow" integer inverse DCT (LL&M, 13-bit constants, 2 extra bits after the column pass), "fancy" triangle
// upsampling of 4:2:2 / 4:2:0 chroma (replication when the chroma plane is at most two samples wide), the 16-bit fixed-point
// YCbCr -> RGB conversion; four-component files (CMYK as Adobe transform 0 / no Adobe marker, YCCK as transform 2; first component at 1 x 1, 2 x 1 or 2 x 2,
// the others at 1 x 1) through the library's YCCK -> CMYK step, the inversion Pillow's plugin applies to every CMYK JPEG and convert('RGB')'s
// nk - MULDIV255(c, nk) arithmetic.  The algorithms are restated from the JPEG standard (ITU T.81 Annex F: Huffman procedures) and from the
// published arithmetic of those libjpeg routines; no source of either library is in this tree.
//
// What makes it a device decoder rather than a device back end is the entropy decoder.  A Huffman-coded scan is one serial bit
// stream per restart interval (most files: per image), but such streams self-synchronise: a decoder started at a wrong position
// or in a wrong state (block of the MCU, coefficient index) falls into step with the true symbol sequence after a few symbols.
// So (after Weissenberger & Schmidt, "Massively parallel Huffman decoding on GPUs", ICPP 2018, and their JPEG follow-up):
//   1. the interval is cut into 128-byte pieces; thread t decodes piece t from its first bit in state (block 0, DC) up to the first
//      symbol boundary at or beyond the end of the piece and records that exit state (bit position, block in MCU, coefficient index);
//   2. rounds: thread t re-decodes its piece from the exit state of thread t - 1; the first piece is right by construction,
//      correctness moves at least one piece per round and in practice everywhere within two or three; the rounds stop when one
//      changes nothing (a fixed point IS the serial decode: every piece starts where its predecessor really ends);
//   3. the numbers of blocks completed per piece are prefix-summed: every piece knows its first block;
//   4. a last decode writes the coefficients (DC differences for now) into the zeroed coefficient array;
//   5. DC prediction = a prefix sum per (interval, component); 6. dequantisation + inverse DCT per block; 7. upsampling + colour
//      conversion per pixel.
// Host work: header parsing, and one pass over the entropy-coded bytes that removes the 0xFF00 stuffing and cuts at restart markers
// (it has to find the end of the scan anyway).  Everything is sized per call in the caller's workspace; a whole list of images is
// decoded by the same launches (blockIdx.y or a binary search over the descriptor tables selects the image).
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gandtr_hip.h"
#include "gdt_common.h"

namespace {

constexpr unsigned SUB_BITS = 1024;          // bits per piece of the parallel entropy decoder
constexpr size_t ALIGN = 256;

// zigzag index -> natural (row-major) index
__constant__ unsigned char d_nat[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                         6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                         39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
const unsigned char h_nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------------ device-side descriptors
struct DImg {
    int width, height, ncomp, B;                   // B: blocks per MCU
    int mcus_x, mcus_y, hs0, vs0;                  // sampling factors of component 0 (the others are 1 x 1)
    int pitch[4], dw[4], dh[4];                    // plane pitch in bytes; real (un-padded) sample columns / rows of the component
    int tdc[4], tac[4];                            // Huffman table numbers (0 / 1) per component
    unsigned total_blocks, first_seg, nseg;
    int ycck;                                      // four components: 1 = YCCK (the first three go through the YCbCr -> RGB tables first), 0 = CMYK
    unsigned long long coef_off;                   // int16 elements from the coefficient base
    unsigned long long plane_off[4];               // bytes from the plane base
    const unsigned char* scan;
    unsigned char* dst;
};
struct DSeg {
    unsigned img, bit0, bit1;                      // bit range of the interval within the image's scan
    unsigned first_block, nblocks;                 // scan-order block numbers within the image
    unsigned first_sub, nsub, sub_bits;
};
// A DHT segment's 16 code-length counts must describe a prefix code (the Kraft check the library makes before it builds its look-up tables):
// with codes assigned in order, the codes of length l must fit l bits and none may be all ones (T.81 C.2; the library refuses such a file, and so
// does the reference's loader).  An over-subscribed table (e.g. three codes of length 1) would index past every table built from it.
inline bool huff_lengths_valid(const unsigned char* bits /* [17], [0] unused */) {
    long long code = 0;
    for (int l = 1; l <= 16; ++l) {
        code += bits[l];
        if (code >= (1LL << l)) return false;
        code <<= 1;
    }
    return true;
}

struct DTab {
    unsigned short look[4][512];                   // 9-bit prefix -> (length << 8 | symbol), 0 = longer code
    int maxcode[4][18];                            // largest code of each length (-1: none)
    int valoff[4][18];                             // index of the symbol of code c of length l = c + valoff[l]
    unsigned char vals[4][256];
    unsigned short quant[4][64];                   // per component, natural order
};

struct HState { unsigned p; int b, z; };
__device__ __forceinline__ unsigned long long pack_state(const HState& s) { return ((unsigned long long)s.p << 16) | ((unsigned)s.b << 8) | (unsigned)s.z; }
__device__ __forceinline__ HState unpack_state(unsigned long long v) { HState s; s.p = (unsigned)(v >> 16); s.b = (int)((v >> 8) & 255); s.z = (int)(v & 255); return s; }

// Decodes symbols from state s until the bit position reaches `limit` (a symbol -- code plus its value bits -- is never split).
// Returns the number of blocks completed.  WRITE: stores the non-zero coefficients of blocks g, g + 1, ... (< g_end) of the image.
// The bit window lives in a 64-bit register; it is topped up by one aligned dword per symbol at most, fetched one refill AHEAD (its
// address does not depend on the code lengths), so the only memory access on the symbol-to-symbol critical path is the table look-up
// (LDS when the whole workgroup works on one file, see the kernels).
template <bool WRITE, typename Tab>
__device__ __forceinline__ unsigned decode_span(const DImg& im, const Tab& tb, unsigned limit, HState& s, short* __restrict__ coef, unsigned g, unsigned g_end) {
    const unsigned* __restrict__ words = (const unsigned*)im.scan;
    const int nY = im.ncomp == 1 ? 1 : im.hs0 * im.vs0, B = im.B;
    const unsigned tsel = (unsigned)(im.tdc[0] | (im.tdc[1] << 1) | (im.tdc[2] << 2) | (im.tdc[3] << 3) | (im.tac[0] << 4) | (im.tac[1] << 5) | (im.tac[2] << 6) | (im.tac[3] << 7));
    unsigned done = 0, p = s.p;
    int b = s.b, z = s.z;
    unsigned next = (p >> 5) + 2;
    unsigned long long buf = (((unsigned long long)__builtin_bswap32(words[next - 2]) << 32) | __builtin_bswap32(words[next - 1])) << (p & 31u);
    int cnt = 64 - (int)(p & 31u);                   // valid bits at the top of buf (> 32 before every symbol)
    unsigned pre = words[next];
    while (p < limit) {
        const unsigned w = (unsigned)(buf >> 32);
        const int c = b < nY ? 0 : b - nY + 1;
        const int t = z == 0 ? (int)((tsel >> c) & 1u) : 2 + (int)((tsel >> (4 + c)) & 1u);
        const unsigned v = w >> 16;
        const unsigned lk = tb.look[t][v >> 7];
        int len, sym;
        if (lk) { len = (int)(lk >> 8); sym = (int)(lk & 255u); }
        else {
            int code = 0;
            for (len = 10; len <= 16; ++len) { code = (int)(v >> (16 - len)); if (code <= tb.maxcode[t][len]) break; }
            if (len > 16) { len = 16; sym = 0; }                         // no such code: only reachable from a wrong guess (or a corrupt file)
            else sym = tb.vals[t][(code + tb.valoff[t][len]) & 255];
        }
        const int ss = sym & 15;
        int val = 0;
        if (ss) {
            const int bits = (int)((w << len) >> (32 - ss));
            val = bits < (1 << (ss - 1)) ? bits - (1 << ss) + 1 : bits;
        }
        int used = len + ss;
        if (z == 0) {
            if (WRITE && val) coef[(size_t)g * 64] = (short)val;
            z = 1;
        } else if (ss) {
            z += sym >> 4;
            if (WRITE && z < 64) coef[(size_t)g * 64 + d_nat[z]] = (short)val;
            z += 1;
        } else {
            used = len;
            z = (sym >> 4) == 15 ? z + 16 : 64;
        }
        p += used; buf <<= used; cnt -= used;
        if (cnt <= 32) {
            buf |= (unsigned long long)__builtin_bswap32(pre) << (32 - cnt);
            cnt += 32;
            pre = words[++next];
        }
        if (z >= 64) {
            z = 0; b = b + 1 == B ? 0 : b + 1;
            ++done; ++g;
            if (WRITE && g >= g_end) break;
        }
    }
    s.p = p; s.b = b; s.z = z;
    return done;
}

// the Huffman tables of the file a workgroup works on, in LDS (the quantisation tables are not needed here)
struct LdsTab {
    unsigned short look[4][512];
    int maxcode[4][18];
    int valoff[4][18];
    unsigned char vals[4][256];
};
static_assert(sizeof(LdsTab) == offsetof(DTab, quant), "LdsTab is the head of DTab");

__device__ __forceinline__ unsigned seg_of_sub(const DSeg* __restrict__ segs, unsigned nseg, unsigned t) {
    unsigned lo = 0, hi = nseg;                    // last segment with first_sub <= t
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (segs[mid].first_sub <= t) lo = mid; else hi = mid; }
    return lo;
}

// which file do the pieces of this workgroup belong to?  (pieces are numbered file by file: first == last => all)
__device__ __forceinline__ int stage_tables(const DSeg* __restrict__ segs, const DTab* __restrict__ tabs, unsigned nseg, unsigned nsub, LdsTab* sh) {
    const unsigned t0 = blockIdx.x * 256u, t1 = min(t0 + 255u, nsub - 1u);
    const unsigned i0 = segs[seg_of_sub(segs, nseg, t0)].img, i1 = segs[seg_of_sub(segs, nseg, t1)].img;
    if (i0 != i1) return -1;
    const unsigned* src = (const unsigned*)&tabs[i0];
    for (unsigned i = threadIdx.x; i < sizeof(LdsTab) / 4; i += 256) ((unsigned*)sh)[i] = src[i];
    __syncthreads();
    return (int)i0;
}

// steps 1 / 2: FIRST = guess pass
template <bool FIRST>
__global__ __launch_bounds__(256) void jpeg_sync_kernel(const DImg* __restrict__ imgs, const DSeg* __restrict__ segs, const DTab* __restrict__ tabs,
                                                        unsigned nseg, unsigned nsub, unsigned long long* __restrict__ exit_state,
                                                        unsigned* __restrict__ nblk, unsigned* __restrict__ changed) {
    __shared__ LdsTab sh;
    const int staged = stage_tables(segs, tabs, nseg, nsub, &sh);
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nsub) return;
    const DSeg sg = segs[seg_of_sub(segs, nseg, t)];
    const unsigned k = t - sg.first_sub;
    const unsigned start = sg.bit0 + k * sg.sub_bits;
    const unsigned limit = min(start + sg.sub_bits, sg.bit1);
    HState s;
    if (FIRST || k == 0) { s.p = start; s.b = 0; s.z = 0; }
    else s = unpack_state(exit_state[t - 1]);
    unsigned done = 0;
    if (s.p < limit) {
        if (staged >= 0) done = decode_span<false>(imgs[sg.img], sh, limit, s, nullptr, 0, 0);
        else done = decode_span<false>(imgs[sg.img], tabs[sg.img], limit, s, nullptr, 0, 0);
    }
    const unsigned long long e = pack_state(s);
    if (!FIRST && (e != exit_state[t] || done != nblk[t])) *changed = 1u;
    exit_state[t] = e;
    nblk[t] = done;
}

// step 3: exclusive prefix sum of the block counts within each interval (one workgroup per interval)
__global__ __launch_bounds__(256) void jpeg_scan_kernel(const DSeg* __restrict__ segs, const unsigned* __restrict__ nblk, unsigned* __restrict__ blk0) {
    __shared__ unsigned sh[256];
    const DSeg sg = segs[blockIdx.x];
    unsigned carry = 0;
    for (unsigned base = 0; base < sg.nsub; base += 256) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = i < sg.nsub ? nblk[sg.first_sub + i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned d = 1; d < 256; d <<= 1) {
            const unsigned add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < sg.nsub) blk0[sg.first_sub + i] = carry + sh[threadIdx.x] - v;
        carry += sh[255];
        __syncthreads();
    }
}

// step 4
__global__ __launch_bounds__(256) void jpeg_write_kernel(const DImg* __restrict__ imgs, const DSeg* __restrict__ segs, const DTab* __restrict__ tabs,
                                                         unsigned nseg, unsigned nsub, const unsigned long long* __restrict__ exit_state,
                                                         const unsigned* __restrict__ blk0, short* __restrict__ coef_base) {
    __shared__ LdsTab sh;
    const int staged = stage_tables(segs, tabs, nseg, nsub, &sh);
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nsub) return;
    const DSeg sg = segs[seg_of_sub(segs, nseg, t)];
    const unsigned k = t - sg.first_sub;
    const unsigned start = sg.bit0 + k * sg.sub_bits;
    const unsigned limit = min(start + sg.sub_bits, sg.bit1);
    HState s;
    if (k == 0) { s.p = start; s.b = 0; s.z = 0; }
    else s = unpack_state(exit_state[t - 1]);
    const DImg& im = imgs[sg.img];
    const unsigned g = sg.first_block + (k == 0 ? 0u : blk0[t]), g_end = sg.first_block + sg.nblocks;
    if (s.p < limit && g < g_end) {
        if (staged >= 0) decode_span<true>(im, sh, limit, s, coef_base + im.coef_off, g, g_end);
        else decode_span<true>(im, tabs[sg.img], limit, s, coef_base + im.coef_off, g, g_end);
    }
}

// step 5: DC prediction, one workgroup per (interval, component)
__global__ __launch_bounds__(256) void jpeg_dc_kernel(const DImg* __restrict__ imgs, const DSeg* __restrict__ segs, short* __restrict__ coef_base) {
    __shared__ int sh[256];
    const DSeg sg = segs[blockIdx.x];
    const DImg& im = imgs[sg.img];
    const int c = blockIdx.y;
    if (c >= im.ncomp) return;
    const unsigned nY = im.ncomp == 1 ? 1u : (unsigned)(im.hs0 * im.vs0);
    const unsigned cnt = c == 0 ? nY : 1u, boff = c == 0 ? 0u : nY + (unsigned)c - 1u;
    const unsigned n = sg.nblocks / (unsigned)im.B * cnt;
    short* coef = coef_base + im.coef_off;
    auto at = [&](unsigned i) -> size_t { return ((size_t)sg.first_block + (size_t)(i / cnt) * im.B + boff + i % cnt) * 64; };
    const unsigned chunk = (n + 255u) / 256u;
    const unsigned lo = min(threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    int sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += coef[at(i)];
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 256; d <<= 1) {
        const int add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - sum;
    for (unsigned i = lo; i < hi; ++i) { run += coef[at(i)]; coef[at(i)] = (short)run; }
}

// step 6: dequantisation + the "islow" inverse DCT (two passes of the Loeffler-Ligtenberg-Moschytz butterfly in 32-bit integers:
// 13-bit constants, the column pass keeps 2 extra bits, final descale by 2^18 with the level shift and the 10-bit wrap-around range limit)
__device__ __forceinline__ void idct_1d(const int in[8], int out[8], const int shift) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    out[0] = (tmp10 + tmp3 + rnd) >> shift; out[7] = (tmp10 - tmp3 + rnd) >> shift;
    out[1] = (tmp11 + tmp2 + rnd) >> shift; out[6] = (tmp11 - tmp2 + rnd) >> shift;
    out[2] = (tmp12 + tmp1 + rnd) >> shift; out[5] = (tmp12 - tmp1 + rnd) >> shift;
    out[3] = (tmp13 + tmp0 + rnd) >> shift; out[4] = (tmp13 - tmp0 + rnd) >> shift;
}

__device__ __forceinline__ unsigned range_limit(int x) {          // the level-shifted 10-bit wrap-around table of the IDCT output stage
    const int i = x & 1023;
    return (unsigned)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}

__global__ __launch_bounds__(64) void jpeg_idct_kernel(const DImg* __restrict__ imgs, const DTab* __restrict__ tabs, const short* __restrict__ coef_base,
                                                       unsigned char* __restrict__ plane_base) {
    const DImg& im = imgs[blockIdx.y];
    const unsigned g = blockIdx.x * 64u + threadIdx.x;
    if (g >= im.total_blocks) return;
    const unsigned mcu = g / (unsigned)im.B, b = g - mcu * (unsigned)im.B;
    const unsigned nY = im.ncomp == 1 ? 1u : (unsigned)(im.hs0 * im.vs0);
    const int c = b < nY ? 0 : (int)(b - nY + 1);
    const int hs = c == 0 && im.ncomp >= 3 ? im.hs0 : 1, vs = c == 0 && im.ncomp >= 3 ? im.vs0 : 1;
    const int bx = (int)(mcu % (unsigned)im.mcus_x) * hs + (c == 0 ? (int)(b % (unsigned)hs) : 0);
    const int by = (int)(mcu / (unsigned)im.mcus_x) * vs + (c == 0 ? (int)(b / (unsigned)hs) : 0);
    const short* __restrict__ cf = coef_base + im.coef_off + (size_t)g * 64;
    const unsigned short* __restrict__ q = tabs[blockIdx.y].quant[c];
    int ws[64];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)cf[r * 8 + col] * (int)q[r * 8 + col];
        idct_1d(in, out, 11);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
    }
    unsigned char* dst = plane_base + im.plane_off[c] + (size_t)(by * 8) * im.pitch[c] + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int out[8];
        idct_1d(ws + r * 8, out, 18);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo |= range_limit(out[k]) << (8 * k); hi |= range_limit(out[4 + k]) << (8 * k); }
        *(uint2*)(dst + (size_t)r * im.pitch[c]) = make_uint2(lo, hi);
    }
}

// step 7: chroma upsampling (triangle filter: 3/4 nearer + 1/4 farther sample, per direction, with the library's alternating rounding
// constants and edge replication) and YCbCr -> RGB in 16-bit fixed point
__device__ __forceinline__ int chroma_at(const DImg& im, const unsigned char* __restrict__ pl, int c, int x, int y) {
    const int pitch = im.pitch[c], dw = im.dw[c], dh = im.dh[c];
    if (im.hs0 == 1) return pl[(size_t)y * pitch + x];
    const int i = x >> 1;
    if (im.vs0 == 1) {                                        // 4:2:2
        const unsigned char* in = pl + (size_t)y * pitch;
        if (dw <= 2) return in[i];
        if (x & 1) return i == dw - 1 ? in[i] : (3 * in[i] + in[i + 1] + 2) >> 2;
        return i == 0 ? in[0] : (3 * in[i] + in[i - 1] + 1) >> 2;
    }
    const int r = y >> 1;                                     // 4:2:0
    if (dw <= 2) return pl[(size_t)r * pitch + i];
    const int r1 = (y & 1) ? min(r + 1, dh - 1) : max(r - 1, 0);
    const unsigned char* in0 = pl + (size_t)r * pitch;
    const unsigned char* in1 = pl + (size_t)r1 * pitch;
    const int cur = 3 * in0[i] + in1[i];
    if (x & 1) return i == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + 3 * in0[i + 1] + in1[i + 1] + 7) >> 4;
    return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + 3 * in0[i - 1] + in1[i - 1] + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const DImg* __restrict__ imgs, const unsigned char* __restrict__ plane_base) {
    const DImg& im = imgs[blockIdx.y];
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)im.width * (unsigned)im.height) return;
    const int y = (int)(i / (unsigned)im.width), x = (int)(i - (unsigned)y * (unsigned)im.width);
    const int yy = plane_base[im.plane_off[0] + (size_t)y * im.pitch[0] + x];
    int r = yy, g = yy, b = yy;
    if (im.ncomp == 3) {
        const int cb = chroma_at(im, plane_base + im.plane_off[1], 1, x, y) - 128, cr = chroma_at(im, plane_base + im.plane_off[2], 2, x, y) - 128;
        r = yy + ((91881 * cr + 32768) >> 16);
        g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        b = yy + ((116130 * cb + 32768) >> 16);
        r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
    } else if (im.ncomp == 4) {
        // Four components.  The library's output is CMYK: the samples as stored (Adobe transform 0 / no Adobe marker) or, for YCCK, (255 - R, 255 - G, 255 - B, K)
        // of the first three components' YCbCr -> RGB (jdcolor.c ycck_cmyk_convert).  Pillow's plugin takes every CMYK JPEG as inverted ("CMYK;I": P = 255 - S) and
        // convert('RGB') computes  nk = 255 - P_k,  out = clip(nk - MULDIV255(P_c, nk))  with  MULDIV255(a, b) = ((t = a * b + 128) + (t >> 8)) >> 8
        int s0 = yy;
        int s1 = chroma_at(im, plane_base + im.plane_off[1], 1, x, y), s2 = chroma_at(im, plane_base + im.plane_off[2], 2, x, y);
        const int nk = chroma_at(im, plane_base + im.plane_off[3], 3, x, y);
        if (im.ycck) {
            const int cb = s1 - 128, cr = s2 - 128;
            const int rr = min(max(yy + ((91881 * cr + 32768) >> 16), 0), 255), gg = min(max(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0), 255);
            const int bb = min(max(yy + ((116130 * cb + 32768) >> 16), 0), 255);
            s0 = 255 - rr; s1 = 255 - gg; s2 = 255 - bb;
        }
        auto muldiv255 = [](int a, int bq) -> int { const int t = a * bq + 128; return ((t >> 8) + t) >> 8; };
        r = min(max(nk - muldiv255(255 - s0, nk), 0), 255);
        g = min(max(nk - muldiv255(255 - s1, nk), 0), 255);
        b = min(max(nk - muldiv255(255 - s2, nk), 0), 255);
    }
    unsigned char* o = im.dst + (size_t)i * 3;
    o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b;
}

// ------------------------------------------------------------------------------------------------ host: headers
struct Reader {
    const unsigned char* f; size_t n, pos;
    bool has(size_t k) const { return pos + k <= n; }
    int u8() { return f[pos++]; }
    int u16() { const int v = (f[pos] << 8) | f[pos + 1]; pos += 2; return v; }
};

int fail(const std::string& msg) { gdt_set_error("jpeg: " + msg); return GDT_ERR_INVALID; }

// walks the entropy-coded bytes from `pos`: counts the bytes that remain after un-stuffing, the restart markers, and finds the marker that ends the scan
struct ScanWalk { size_t data_bytes = 0; int restarts = 0; int end_marker = -1; size_t end_pos = 0; bool overflow = false; };
// (dst == nullptr: measure only; otherwise at most `cap` bytes and `max_restarts` interval offsets are written -- a file that does not
// match the info it is extracted with must not overrun the caller's buffers)
ScanWalk walk_scan(const unsigned char* f, size_t n, size_t pos, unsigned char* dst, size_t cap, unsigned int* seg_off, int max_restarts) {
    ScanWalk w;
    size_t o = 0;
    if (seg_off) seg_off[0] = 0;
    while (pos < n) {
        const unsigned char* hit = (const unsigned char*)memchr(f + pos, 0xFF, n - pos);
        const size_t stop = hit ? (size_t)(hit - f) : n;
        if (dst) {
            if (o + (stop - pos) + 1 > cap) { w.overflow = true; break; }
            memcpy(dst + o, f + pos, stop - pos);
        }
        o += stop - pos;
        pos = stop;
        if (pos >= n) break;
        size_t q = pos + 1;
        while (q < n && f[q] == 0xFF) ++q;                    // fill bytes
        if (q >= n) { pos = n; break; }
        const int m = f[q];
        if (m == 0x00) { if (dst) dst[o] = 0xFF; ++o; pos = q + 1; continue; }
        if (m >= 0xD0 && m <= 0xD7) {
            ++w.restarts;
            if (seg_off) { if (w.restarts > max_restarts) { w.overflow = true; break; } seg_off[w.restarts] = (unsigned int)o; }
            pos = q + 1;
            continue;
        }
        w.end_marker = m; w.end_pos = q + 1;
        break;
    }
    if (w.end_marker < 0) w.end_pos = n;
    w.data_bytes = o;
    return w;
}

size_t scan_capacity_for(size_t data_bytes) { return (data_bytes + 15) / 16 * 16 + 32; }

int parse_impl(const unsigned char* f, size_t n, gdt_jpeg_info* info) {
    memset(info, 0, sizeof(*info));
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) return fail("not a JPEG file (no SOI marker)");
    Reader r{f, n, 2};
    bool have_frame = false, have_q[4] = {false, false, false, false}, have_h[4] = {false, false, false, false}, bad_h[4] = {false, false, false, false}, jfif = false, adobe = false;
    int adobe_transform = -1, comp_id[4] = {0, 0, 0, 0};
    for (;;) {
        if (!r.has(2)) return fail("truncated before the scan");
        if (r.u8() != 0xFF) return fail("marker expected");
        int m = r.u8();
        while (m == 0xFF && r.has(1)) m = r.u8();
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return fail("no scan");
        if (!r.has(2)) return fail("truncated segment");
        const int len = r.u16();
        if (len < 2 || !r.has((size_t)len - 2)) return fail("truncated segment");
        const size_t end = r.pos + len - 2;
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (have_frame) return fail("more than one frame");
            info->progressive = m == 0xC2;
            if (len < 8) return fail("bad frame header");
            const int prec = r.u8();
            info->height = r.u16(); info->width = r.u16(); info->ncomp = r.u8();
            if (prec != 8) return fail("only 8-bit samples are decoded on the device");
            if (info->ncomp != 1 && info->ncomp != 3 && info->ncomp != 4) return fail("only files of one, three or four components are decoded on the device");
            if (info->width <= 0 || info->height <= 0) return fail("empty image (or height deferred to a DNL marker)");
            if (len != 8 + 3 * info->ncomp) return fail("bad frame header");
            for (int c = 0; c < info->ncomp; ++c) {
                comp_id[c] = r.u8();
                const int hv = r.u8();
                info->hs[c] = hv >> 4; info->vs[c] = hv & 15; info->tq[c] = r.u8();
                if (info->tq[c] > 3) return fail("bad quantisation table number");
            }
            have_frame = true;
        } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail("lossless / hierarchical / arithmetic-coded files are not decoded on the device");
        } else if (m == 0xCC) {
            return fail("arithmetic-coded files are not decoded on the device");
        } else if (m == 0xC4) {
            while (r.pos < end) {
                if (end - r.pos < 17) return fail("bad Huffman table");
                const int tc_th = r.u8(), tc = tc_th >> 4, th = tc_th & 15;
                if (tc > 1 || th > 3) return fail("bad Huffman table number");
                if (th > 1) {                  // tables 2 / 3: progressive files only (their coefficient decoder reads the tables itself)
                    if (!info->progressive && have_frame) return fail("Huffman table numbers above 1 (not baseline)");
                    int total = 0; unsigned char hb[17] = {0};
                    for (int l = 1; l <= 16; ++l) { hb[l] = (unsigned char)r.u8(); total += hb[l]; }
                    if (total > 256 || end - r.pos < (size_t)total) return fail("bad Huffman table");      // (its lengths are checked by the coefficient decoder, at use)
                    r.pos += total;
                    continue;
                }
                const int idx = tc * 2 + th;
                int total = 0;
                info->huff_bits[idx][0] = 0;
                for (int l = 1; l <= 16; ++l) { info->huff_bits[idx][l] = (unsigned char)r.u8(); total += info->huff_bits[idx][l]; }
                if (total > 256 || end - r.pos < (size_t)total) return fail("bad Huffman table");
                bad_h[idx] = !huff_lengths_valid(info->huff_bits[idx]);
                memset(info->huff_vals[idx], 0, 256);
                for (int i = 0; i < total; ++i) info->huff_vals[idx][i] = (unsigned char)r.u8();
                have_h[idx] = true;
            }
        } else if (m == 0xDB) {
            while (r.pos < end) {
                const int pq_tq = r.u8(), pq = pq_tq >> 4, tq = pq_tq & 15;
                if (tq > 3 || pq > 1 || end - r.pos < (size_t)(64 * (pq + 1))) return fail("bad quantisation table");
                for (int i = 0; i < 64; ++i) info->quant[tq][h_nat[i]] = (unsigned short)(pq ? r.u16() : r.u8());
                have_q[tq] = true;
            }
        } else if (m == 0xDD) {
            if (len != 4) return fail("bad restart interval");
            info->restart_interval = r.u16();
        } else if (m == 0xE0) {
            if (len >= 7 && memcmp(f + r.pos, "JFIF", 5) == 0) jfif = true;
        } else if (m == 0xEE) {
            if (len >= 14 && memcmp(f + r.pos, "Adobe", 5) == 0) { adobe = true; adobe_transform = f[r.pos + 11]; }
        } else if (m == 0xDA && info->progressive) {
            if (!have_frame) return fail("scan before the frame header");
            for (int c = 0; c < info->ncomp; ++c)
                if (!have_q[info->tq[c]]) return fail("frame refers to a quantisation table that was not defined");
            r.pos -= 4;                        // the coefficient decoder starts at this SOS marker (tables may change between the scans)
            break;
        } else if (m == 0xDA) {
            if (!have_frame) return fail("scan before the frame header");
            if (len < 6 + 2) return fail("bad scan header");
            const int ns = r.u8();
            if (ns != info->ncomp) return fail("only single-scan (interleaved) files are decoded on the device");
            if (len != 6 + 2 * ns) return fail("bad scan header");
            for (int c = 0; c < ns; ++c) {
                const int id = r.u8(), tt = r.u8();
                if (id != comp_id[c]) return fail("scan components out of frame order");
                info->td[c] = tt >> 4; info->ta[c] = tt & 15;
                if (info->td[c] > 1 || info->ta[c] > 1) return fail("Huffman table numbers above 1 (not baseline)");
                if (!have_h[info->td[c]] || !have_h[2 + info->ta[c]]) return fail("scan refers to a Huffman table that was not defined");
                if (bad_h[info->td[c]] || bad_h[2 + info->ta[c]]) return fail("bad Huffman table (its code lengths are not a prefix code)");
                if (!have_q[info->tq[c]]) return fail("frame refers to a quantisation table that was not defined");
            }
            const int ss = r.u8(), se = r.u8(), ahal = r.u8();
            if (ss != 0 || se != 63 || ahal != 0) return fail("not a sequential scan");
            r.pos = end;
            break;
        }
        r.pos = end;
    }
    // colour space (the library's rule: JFIF => YCbCr; Adobe => by its transform flag; otherwise by the component ids)
    if (info->ncomp == 3) {
        if (!jfif && adobe && adobe_transform != 1) return fail("RGB-coded (Adobe transform 0) files are not decoded on the device");
        if (!jfif && !adobe && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') return fail("RGB-coded files are not decoded on the device");
        const bool chroma_ok = info->hs[1] == 1 && info->vs[1] == 1 && info->hs[2] == 1 && info->vs[2] == 1;
        const bool luma_ok = (info->hs[0] == 1 && info->vs[0] == 1) || (info->hs[0] == 2 && info->vs[0] == 1) || (info->hs[0] == 2 && info->vs[0] == 2);
        if (!chroma_ok || !luma_ok) return fail("sampling factors other than 4:4:4, 4:2:2 and 4:2:0 are not decoded on the device");
    } else if (info->ncomp == 4) {
        // CMYK (Adobe transform 0, or no Adobe marker: the library's guess for four components) or YCCK (transform 2)
        if (adobe && adobe_transform != 0 && adobe_transform != 2) return fail("four-component file with an unknown Adobe transform");
        info->adobe_transform = adobe && adobe_transform == 2 ? 2 : 0;
        bool rest_ok = true;
        for (int c = 1; c < 4; ++c) rest_ok = rest_ok && info->hs[c] == 1 && info->vs[c] == 1;
        const bool first_ok = (info->hs[0] == 1 && info->vs[0] == 1) || (info->hs[0] == 2 && info->vs[0] == 1) || (info->hs[0] == 2 && info->vs[0] == 2);
        if (!rest_ok || !first_ok) return fail("four-component files are decoded on the device with components 2-4 at 1 x 1 and the first at 1 x 1, 2 x 1 or 2 x 2");
    } else {
        info->hs[0] = 1; info->vs[0] = 1;                   // a single-component scan is never interleaved: one block per MCU
    }
    const int hmax = info->hs[0], vmax = info->vs[0];
    info->mcus_x = (info->width + 8 * hmax - 1) / (8 * hmax);
    info->mcus_y = (info->height + 8 * vmax - 1) / (8 * vmax);
    info->blocks_per_mcu = info->ncomp == 1 ? 1 : hmax * vmax + info->ncomp - 1;
    if ((long long)info->mcus_x * info->mcus_y * info->blocks_per_mcu >= (1LL << 26)) return fail("image too large");
    info->scan_offset = r.pos;
    if (info->progressive) {                  // (the scans are walked -- and a truncated file refused -- by gdt_jpeg_progressive_coefficients)
        for (int c = 0; c < 4; ++c) info->comp_id[c] = comp_id[c];
        info->nsegments = 1; info->scan_capacity = scan_capacity_for(0);
        return GDT_OK;
    }
    const ScanWalk w = walk_scan(f, n, r.pos, nullptr, 0, nullptr, 0);
    if (w.end_marker >= 0 && w.end_marker != 0xD9) return fail("more than one scan (or a marker inside the scan) -- not decoded on the device");
    // A scan that runs into the end of the buffer is a truncated file: the reference's loader (pil_loader, datahelpers.py:39-47) raises
    // OSError "image file is truncated" for it unless every MCU was already decoded, which the host cannot tell without decoding.  Refused:
    // the caller's host loader (Pillow itself) then raises or tolerates exactly as the reference does.
    if (w.end_marker < 0) return fail("truncated scan (no EOI marker)");
    if (w.data_bytes >= (1u << 28)) return fail("scan too large");
    const long long total_mcus = (long long)info->mcus_x * info->mcus_y;
    const long long expect = info->restart_interval > 0 ? (total_mcus + info->restart_interval - 1) / info->restart_interval : 1;
    if (w.restarts + 1 != expect) return fail("the restart markers do not match the restart interval");
    info->nsegments = w.restarts + 1;
    info->scan_capacity = scan_capacity_for(w.data_bytes);
    return GDT_OK;
}

// ------------------------------------------------------------------------------------------------ host: decode plan
struct Plan {
    std::vector<DImg> imgs;
    std::vector<DSeg> segs;
    std::vector<DTab> tabs;
    unsigned nsub = 0, max_blocks = 0, max_pixels = 0;
    size_t coef_elems = 0, plane_bytes = 0;
    size_t off_imgs = 0, off_segs = 0, off_tabs = 0, off_exit = 0, off_nblk = 0, off_blk0 = 0, off_flags = 0, off_coef = 0, off_planes = 0, total = 0;
};
constexpr unsigned MAX_ROUNDS = 1024;

void build_table(const gdt_jpeg_info& info, int idx, DTab& t) {
    int code = 0, k = 0;
    memset(t.look[idx], 0, sizeof(t.look[idx]));
    memcpy(t.vals[idx], info.huff_vals[idx], 256);
    for (int l = 0; l < 18; ++l) { t.maxcode[idx][l] = -1; t.valoff[idx][l] = 0; }
    for (int l = 1; l <= 16; ++l) {
        const int cnt = info.huff_bits[idx][l];
        t.valoff[idx][l] = k - code;
        for (int i = 0; i < cnt; ++i, ++code, ++k) {
            if (l <= 9 && code < (1 << l)) {
                const int lo = code << (9 - l);
                for (int x = 0; x < (1 << (9 - l)); ++x) t.look[idx][(lo + x) & 511] = (unsigned short)((l << 8) | info.huff_vals[idx][k & 255]);
            }
        }
        t.maxcode[idx][l] = cnt ? code - 1 : -1;
        code <<= 1;
    }
}

int make_plan(const gdt_jpeg_item* items, int n, int mode, Plan& p) {
    GDT_REQUIRE(items != nullptr && n >= 1 && n <= 65535, "jpeg: 1..65535 images per call");
    GDT_REQUIRE(mode == 0 || mode == 1, "jpeg: mode 0 (parallel entropy decoding) or 1 (one thread per restart interval)");
    p.imgs.resize(n); p.tabs.resize(n);
    for (int i = 0; i < n; ++i) {
        const gdt_jpeg_item& it = items[i];
        GDT_REQUIRE(it.info != nullptr && it.seg_off != nullptr, "jpeg: item without info / segment offsets");
        const gdt_jpeg_info& f = *it.info;
        GDT_REQUIRE((f.ncomp == 1 || f.ncomp == 3 || f.ncomp == 4) && f.width > 0 && f.height > 0 && f.nsegments >= 1 && f.mcus_x > 0 && f.mcus_y > 0,
                    "jpeg: info was not filled by gdt_jpeg_parse");
        GDT_REQUIRE(!f.progressive, "jpeg: progressive files are decoded through gdt_jpeg_progressive_coefficients + gdt_jpeg_decode_coef_u8_batch");
        GDT_REQUIRE(((uintptr_t)it.scan & 15) == 0, "jpeg: scan buffers must be 16-byte aligned");
        DImg& d = p.imgs[i];
        memset(&d, 0, sizeof(d));
        d.width = f.width; d.height = f.height; d.ncomp = f.ncomp; d.B = f.blocks_per_mcu;
        d.mcus_x = f.mcus_x; d.mcus_y = f.mcus_y; d.hs0 = f.ncomp >= 3 ? f.hs[0] : 1; d.vs0 = f.ncomp >= 3 ? f.vs[0] : 1; d.ycck = f.ncomp == 4 && f.adobe_transform == 2;
        d.total_blocks = (unsigned)((long long)f.mcus_x * f.mcus_y * f.blocks_per_mcu);
        d.coef_off = p.coef_elems;
        p.coef_elems += (size_t)d.total_blocks * 64;
        for (int c = 0; c < f.ncomp; ++c) {
            const int hs = c == 0 ? d.hs0 : 1, vs = c == 0 ? d.vs0 : 1;
            d.pitch[c] = f.mcus_x * hs * 8;
            d.dw[c] = (f.width * hs + d.hs0 - 1) / d.hs0;
            d.dh[c] = (f.height * vs + d.vs0 - 1) / d.vs0;
            d.tdc[c] = f.td[c]; d.tac[c] = f.ta[c];
            d.plane_off[c] = p.plane_bytes;
            p.plane_bytes += ((size_t)d.pitch[c] * f.mcus_y * vs * 8 + 15) / 16 * 16;
            memcpy(p.tabs[i].quant[c], f.quant[f.tq[c]], 128);
        }
        for (int t = 0; t < 4; ++t) build_table(f, t, p.tabs[i]);
        d.scan = it.scan; d.dst = it.dst_hwc;
        d.first_seg = (unsigned)p.segs.size(); d.nseg = (unsigned)f.nsegments;
        const unsigned mcus = (unsigned)(f.mcus_x * f.mcus_y);
        for (int s = 0; s < f.nsegments; ++s) {
            DSeg g;
            g.img = (unsigned)i;
            GDT_REQUIRE(it.seg_off[s] <= it.seg_off[s + 1] && it.seg_off[s + 1] + 16 <= f.scan_capacity, "jpeg: bad segment offsets");
            g.bit0 = it.seg_off[s] * 8u; g.bit1 = it.seg_off[s + 1] * 8u;
            const unsigned m0 = f.restart_interval > 0 ? (unsigned)s * (unsigned)f.restart_interval : 0u;
            const unsigned m1 = f.restart_interval > 0 ? std::min(mcus, m0 + (unsigned)f.restart_interval) : mcus;
            GDT_REQUIRE(m0 < m1, "jpeg: more restart intervals than MCUs");
            g.first_block = m0 * (unsigned)d.B; g.nblocks = (m1 - m0) * (unsigned)d.B;
            const unsigned bits = g.bit1 - g.bit0;
            g.sub_bits = mode == 1 ? std::max(bits, 1u) : SUB_BITS;
            g.nsub = std::max(1u, (bits + g.sub_bits - 1) / g.sub_bits);
            g.first_sub = p.nsub;
            p.nsub += g.nsub;
            p.segs.push_back(g);
        }
        p.max_blocks = std::max(p.max_blocks, d.total_blocks);
        p.max_pixels = std::max(p.max_pixels, (unsigned)f.width * (unsigned)f.height);
        GDT_REQUIRE((long long)f.width * f.height < (1LL << 31), "jpeg: image too large");
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + ALIGN - 1) / ALIGN * ALIGN; return at; };
    p.off_imgs = take(p.imgs.size() * sizeof(DImg));
    p.off_segs = take(p.segs.size() * sizeof(DSeg));
    p.off_tabs = take(p.tabs.size() * sizeof(DTab));
    p.off_exit = take((size_t)p.nsub * 8);
    p.off_nblk = take((size_t)p.nsub * 4);
    p.off_blk0 = take((size_t)p.nsub * 4);
    p.off_flags = take((size_t)MAX_ROUNDS * 4);
    p.off_coef = take(p.coef_elems * 2);
    p.off_planes = take(p.plane_bytes);
    p.total = o + ALIGN;
    return GDT_OK;
}

int run_decode(const gdt_jpeg_item* items, int n, int mode, const Plan& p, char* ws, hipStream_t stream) {
    for (int i = 0; i < n; ++i) GDT_REQUIRE(items[i].scan != nullptr && items[i].dst_hwc != nullptr, "jpeg: null scan / output buffer");
    DImg* d_imgs = (DImg*)(ws + p.off_imgs);
    DSeg* d_segs = (DSeg*)(ws + p.off_segs);
    DTab* d_tabs = (DTab*)(ws + p.off_tabs);
    unsigned long long* d_exit = (unsigned long long*)(ws + p.off_exit);
    unsigned* d_nblk = (unsigned*)(ws + p.off_nblk);
    unsigned* d_blk0 = (unsigned*)(ws + p.off_blk0);
    unsigned* d_flags = (unsigned*)(ws + p.off_flags);
    short* d_coef = (short*)(ws + p.off_coef);
    unsigned char* d_planes = (unsigned char*)(ws + p.off_planes);
    GDT_CHECK_HIP(hipMemcpyAsync(d_imgs, p.imgs.data(), p.imgs.size() * sizeof(DImg), hipMemcpyHostToDevice, stream));
    GDT_CHECK_HIP(hipMemcpyAsync(d_segs, p.segs.data(), p.segs.size() * sizeof(DSeg), hipMemcpyHostToDevice, stream));
    GDT_CHECK_HIP(hipMemcpyAsync(d_tabs, p.tabs.data(), p.tabs.size() * sizeof(DTab), hipMemcpyHostToDevice, stream));
    GDT_CHECK_HIP(hipStreamSynchronize(stream));                // (the staging vectors live in the caller's frame only)
    GDT_CHECK_HIP(hipMemsetAsync(d_coef, 0, p.coef_elems * 2, stream));
    GDT_CHECK_HIP(hipMemsetAsync(d_flags, 0, (size_t)MAX_ROUNDS * 4, stream));
    const unsigned nseg = (unsigned)p.segs.size(), grid_sub = (p.nsub + 255u) / 256u;
    if (mode == 0) {
        hipLaunchKernelGGL(jpeg_sync_kernel<true>, dim3(grid_sub), dim3(256), 0, stream, d_imgs, d_segs, d_tabs, nseg, p.nsub, d_exit, d_nblk, d_flags);
        unsigned round = 0;
        bool converged = false;
        // the longest interval bounds the number of rounds (one piece per round in the worst case)
        unsigned longest = 1;
        for (const DSeg& g : p.segs) longest = std::max(longest, g.nsub);
        while (!converged && round < MAX_ROUNDS && round < longest + 4) {
            const unsigned group = std::min(4u, MAX_ROUNDS - round);
            for (unsigned k = 0; k < group; ++k)
                hipLaunchKernelGGL(jpeg_sync_kernel<false>, dim3(grid_sub), dim3(256), 0, stream, d_imgs, d_segs, d_tabs, nseg, p.nsub, d_exit, d_nblk,
                                   d_flags + round + k);
            unsigned flags[4] = {1, 1, 1, 1};
            GDT_CHECK_HIP(hipMemcpyAsync(flags, d_flags + round, group * 4, hipMemcpyDeviceToHost, stream));
            GDT_CHECK_HIP(hipStreamSynchronize(stream));
            round += group;
            converged = flags[group - 1] == 0;
        }
        if (!converged) { gdt_set_error("jpeg: the parallel entropy decoder did not reach a fixed point"); return GDT_ERR_NOT_CONVERGED; }
        static const bool trace = getenv("GDT_JPEG_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[jpeg] %d files, %u intervals, %u pieces (longest interval %u): fixed point after %u repair rounds\n", n, nseg, p.nsub, longest, round);
        hipLaunchKernelGGL(jpeg_scan_kernel, dim3(nseg), dim3(256), 0, stream, d_segs, d_nblk, d_blk0);
    } else {
        GDT_CHECK_HIP(hipMemsetAsync(d_blk0, 0, (size_t)p.nsub * 4, stream));
    }
    hipLaunchKernelGGL(jpeg_write_kernel, dim3(grid_sub), dim3(256), 0, stream, d_imgs, d_segs, d_tabs, nseg, p.nsub, d_exit, d_blk0, d_coef);
    hipLaunchKernelGGL(jpeg_dc_kernel, dim3(nseg, 4), dim3(256), 0, stream, d_imgs, d_segs, d_coef);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((p.max_blocks + 63u) / 64u, n), dim3(64), 0, stream, d_imgs, d_tabs, d_coef, d_planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((p.max_pixels + 255u) / 256u, n), dim3(256), 0, stream, d_imgs, d_planes);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}


// ------------------------------------------------------------------------------------------------ host: progressive files (ITU T.81 Annex G)
// A progressive file codes every coefficient in several scans: a band of the zigzag sequence (spectral selection Ss..Se) at a time, and within a
// band first the high bits (successive approximation: "first" scans code value >> Al) and then one more bit per "refinement" scan.  The scans refine
// the same coefficient array one after the other, so this is a sequential pass over the file: it runs on the host and hands the device the finished
// coefficients (gdt_jpeg_decode_coef_u8_batch).  Procedures restated from the standard: G.1.2 (DC first: F.2.2.1 on the point-transformed value;
// DC refinement: one bit per block; AC first: run / size symbols with end-of-band runs EOBn; AC refinement: correction bits for the coefficients that are
// already non-zero, new coefficients of magnitude 1 placed after r ZERO-HISTORY positions).
struct HuffHost {
    int maxcode[18], valptr[17], mincode[17];
    unsigned char vals[256];
    unsigned short look[512];                     // 9-bit prefix -> (length << 8 | symbol), 0 = longer code
    bool defined = false, bad = false;            // bad: defined by a DHT whose lengths are no prefix code -- an error once a scan uses it (as in the library)
    void build(const unsigned char* bits /* [17] */, const unsigned char* v) {
        defined = true;
        bad = !huff_lengths_valid(bits);
        if (bad) return;                              // (with valid lengths every code fits its length: code << (9 - l) stays below 512)
        memcpy(vals, v, 256);
        memset(look, 0, sizeof(look));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++code, ++k)
                if (l <= 9) for (int f = 0; f < (1 << (9 - l)); ++f) look[(code << (9 - l)) | f] = (unsigned short)((l << 8) | v[k]);
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct BitReader {
    const unsigned char* f; size_t n, pos;
    unsigned long long acc = 0; int cnt = 0;
    int marker = -1;                              // a marker was met: zeros are fed from here on
    bool starved = false;                         // ... or the buffer ended
    void fill() {
        while (cnt <= 56) {
            unsigned b = 0;
            if (marker < 0 && pos < n) {
                b = f[pos++];
                if (b == 0xFF) {
                    size_t q = pos;
                    while (q < n && f[q] == 0xFF) ++q;
                    if (q >= n) { starved = true; pos = n; b = 0; }
                    else if (f[q] == 0x00) pos = q + 1;                        // stuffed byte
                    else { marker = f[q]; pos = q + 1; b = 0; }
                }
            } else if (marker < 0) starved = true;
            acc |= (unsigned long long)b << (56 - cnt);
            cnt += 8;
        }
    }
    unsigned peek(int k) { if (cnt < k) fill(); return (unsigned)(acc >> (64 - k)); }
    void skip(int k) { acc <<= k; cnt -= k; }
    unsigned bits(int k) { if (k == 0) return 0; const unsigned v = peek(k); skip(k); return v; }
    int decode(const HuffHost& h) {
        const unsigned p = peek(16);
        const unsigned short e = h.look[p >> 7];
        if (e) { skip(e >> 8); return e & 255; }
        int code = (int)(p >> 6), l = 10;
        while (l <= 16 && code > h.maxcode[l]) { ++l; code = (int)(p >> (16 - l)); }
        if (l > 16) { skip(16); return 0; }       // corrupt stream: a zero symbol keeps the walk bounded
        skip(l);
        return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
    }
    // restart: drop the rest of the current byte; the RSTn marker has been (or is about to be) consumed
    void restart() {
        acc = 0; cnt = 0;
        if (marker < 0) {                         // the marker lies ahead (the interval ended exactly on a byte boundary and nothing was pre-fetched past it)
            size_t q = pos;
            while (q + 1 < n && !(f[q] == 0xFF && f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7)) {
                if (f[q] == 0xFF && f[q + 1] != 0x00 && f[q + 1] != 0xFF) break;
                ++q;
            }
            if (q + 1 < n && f[q] == 0xFF && f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7) pos = q + 2;
        } else if (marker >= 0xD0 && marker <= 0xD7) marker = -1;
    }
};
inline int extend(unsigned v, int s) { return s == 0 ? 0 : ((int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v); }

int progressive_impl(const unsigned char* f, size_t n, const gdt_jpeg_info& info, short* coef) {
    if (!info.progressive || info.scan_offset < 2 || info.scan_offset >= n) return fail("info was not filled by gdt_jpeg_parse for a progressive file");
    const int ncomp = info.ncomp, B = info.blocks_per_mcu;
    const int hmax = ncomp >= 3 ? info.hs[0] : 1, vmax = ncomp >= 3 ? info.vs[0] : 1;
    const long long total_blocks = (long long)info.mcus_x * info.mcus_y * B;
    memset(coef, 0, (size_t)total_blocks * 64 * sizeof(short));
    const int nY = ncomp == 1 ? 1 : hmax * vmax;
    // the block grid a NON-interleaved scan of component c covers (its own size, not padded to whole MCUs) and its block -> storage index
    auto comp_w = [&](int c) { const int hs = (ncomp >= 3 && c == 0) ? hmax : 1; return ((info.width * hs + hmax - 1) / hmax + 7) / 8; };
    auto comp_h = [&](int c) { const int vs = (ncomp >= 3 && c == 0) ? vmax : 1; return ((info.height * vs + vmax - 1) / vmax + 7) / 8; };
    auto block_at = [&](int c, int bx, int by) -> short* {
        long long g;
        if (ncomp >= 3 && c == 0) g = ((long long)(by / vmax) * info.mcus_x + bx / hmax) * B + (by % vmax) * hmax + (bx % hmax);
        else if (ncomp >= 3) g = ((long long)by * info.mcus_x + bx) * B + nY + (c - 1);
        else g = (long long)by * info.mcus_x + bx;
        return coef + g * 64;
    };
    HuffHost dc[4], ac[4];
    for (int t = 0; t < 2; ++t) {                 // tables 0 / 1 as the headers before the first scan left them
        int any = 0;
        for (int l = 1; l <= 16; ++l) any += info.huff_bits[t][l];
        if (any) dc[t].build(info.huff_bits[t], info.huff_vals[t]);
        any = 0;
        for (int l = 1; l <= 16; ++l) any += info.huff_bits[2 + t][l];
        if (any) ac[t].build(info.huff_bits[2 + t], info.huff_vals[2 + t]);
    }
    int restart_interval = info.restart_interval;
    // tables 2 / 3 defined BEFORE the first scan were skipped by the header parser: re-read every DHT of the file up to the first scan
    {
        size_t pos = 2;
        while (pos + 4 <= info.scan_offset) {
            if (f[pos] != 0xFF) break;
            const int m = f[pos + 1];
            if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0xFF) { pos += (m == 0xFF) ? 1 : 2; continue; }
            const size_t len = ((size_t)f[pos + 2] << 8) | f[pos + 3];
            if (len < 2 || pos + 2 + len > n) break;
            if (m == 0xC4) {
                size_t q = pos + 4; const size_t end = pos + 2 + len;
                while (q + 17 <= end) {
                    const int tc = f[q] >> 4, th = f[q] & 15;
                    int total = 0; unsigned char bits[17] = {0}, vals[256] = {0};
                    for (int l = 1; l <= 16; ++l) { bits[l] = f[q + l]; total += bits[l]; }
                    if (total > 256 || q + 17 + total > end || tc > 1 || th > 3) break;
                    memcpy(vals, f + q + 17, total);
                    (tc ? ac[th] : dc[th]).build(bits, vals);
                    q += 17 + total;
                }
            }
            pos += 2 + len;
        }
    }
    size_t pos = info.scan_offset;
    bool saw_eoi = false;
    int scans = 0;
    while (pos + 2 <= n) {
        if (f[pos] != 0xFF) return fail("progressive: marker expected between the scans");
        int m = f[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD9) { saw_eoi = true; break; }
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) break;
        const size_t len = ((size_t)f[pos] << 8) | f[pos + 1];
        if (len < 2 || pos + len > n) return fail("progressive: truncated segment");
        const size_t end = pos + len;
        if (m == 0xC4) {
            size_t q = pos + 2;
            while (q < end) {
                if (end - q < 17) return fail("bad Huffman table");
                const int tc = f[q] >> 4, th = f[q] & 15;
                int total = 0; unsigned char bits[17] = {0}, vals[256] = {0};
                for (int l = 1; l <= 16; ++l) { bits[l] = f[q + l]; total += bits[l]; }
                if (tc > 1 || th > 3 || total > 256 || end - q - 17 < (size_t)total) return fail("bad Huffman table");
                memcpy(vals, f + q + 17, total);
                (tc ? ac[th] : dc[th]).build(bits, vals);
                q += 17 + total;
            }
            pos = end; continue;
        }
        if (m == 0xDD) { if (len != 4) return fail("bad restart interval"); restart_interval = (f[pos + 2] << 8) | f[pos + 3]; pos = end; continue; }
        if (m == 0xDB) return fail("progressive: quantisation tables redefined between the scans are not supported");
        if (m != 0xDA) { pos = end; continue; }    // (APPn / COM between scans)
        // ---- one scan
        if (len < 8) return fail("bad scan header");
        const int ns = f[pos + 2];
        if (ns < 1 || ns > ncomp || len != (size_t)(6 + 2 * ns)) return fail("bad scan header");
        int sc[4], td[4], ta[4];
        for (int i = 0; i < ns; ++i) {
            const int id = f[pos + 3 + 2 * i], tt = f[pos + 4 + 2 * i];
            int c = -1;
            for (int k = 0; k < ncomp; ++k) if (info.comp_id[k] == id) c = k;
            if (c < 0 || (i > 0 && c <= sc[i - 1])) return fail("progressive: scan names an unknown component (or components out of order)");
            sc[i] = c; td[i] = tt >> 4; ta[i] = tt & 15;
            if (td[i] > 3 || ta[i] > 3) return fail("bad Huffman table number");
        }
        const int Ss = f[pos + 3 + 2 * ns], Se = f[pos + 4 + 2 * ns], Ah = f[pos + 5 + 2 * ns] >> 4, Al = f[pos + 5 + 2 * ns] & 15;
        if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1))
            return fail("progressive: bad spectral selection / successive approximation parameters");
        for (int i = 0; i < ns; ++i) {
            if (Ss == 0 && Ah == 0 && !dc[td[i]].defined) return fail("scan refers to a Huffman table that was not defined");
            if (Ss > 0 && !ac[ta[i]].defined) return fail("scan refers to a Huffman table that was not defined");
            if ((Ss == 0 && Ah == 0 && dc[td[i]].bad) || (Ss > 0 && ac[ta[i]].bad)) return fail("bad Huffman table (its code lengths are not a prefix code)");
        }
        BitReader br{f, n, end};
        ++scans;
        const bool interleaved = ns > 1;
        // geometry of the scan: MCUs of the frame (interleaved) or the component's own blocks
        const int sw = interleaved ? info.mcus_x : comp_w(sc[0]), sh = interleaved ? info.mcus_y : comp_h(sc[0]);
        const long long units = (long long)sw * sh;
        int pred[4] = {0, 0, 0, 0};
        unsigned eobrun = 0;
        const int p1 = 1 << Al, m1 = -(1 << Al);
        long long until_restart = restart_interval > 0 ? restart_interval : -1;
        for (long long u = 0; u < units; ++u) {
            if (until_restart == 0) {
                br.restart();
                pred[0] = pred[1] = pred[2] = pred[3] = 0; eobrun = 0;
                until_restart = restart_interval;
            }
            if (until_restart > 0) --until_restart;
            const int ux = (int)(u % sw), uy = (int)(u / sw);
            for (int i = 0; i < ns; ++i) {
                const int c = sc[i];
                const int bw = interleaved && ncomp >= 3 && c == 0 ? hmax : 1, bh = interleaved && ncomp >= 3 && c == 0 ? vmax : 1;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        short* blk = interleaved ? block_at(c, ux * bw + bx, uy * bh + by) : block_at(c, ux, uy);
                        if (Ss == 0) {
                            if (Ah == 0) {                        // DC, first scan
                                const int ssz = br.decode(dc[td[i]]) & 15;
                                pred[c] += extend(br.bits(ssz), ssz);
                                blk[0] = (short)(pred[c] * (1 << Al));
                            } else if (br.bits(1)) blk[0] = (short)(blk[0] | p1);     // DC refinement
                        } else if (Ah == 0) {                     // AC band, first scan
                            if (eobrun > 0) { --eobrun; continue; }
                            for (int k = Ss; k <= Se; ++k) {
                                const int rs = br.decode(ac[ta[i]]), r = rs >> 4, sz = rs & 15;
                                if (sz) {
                                    k += r;
                                    if (k > 63) break;
                                    blk[h_nat[k]] = (short)(extend(br.bits(sz), sz) * (1 << Al));
                                } else if (r == 15) k += 15;
                                else { eobrun = (1u << r) + (r ? br.bits(r) : 0u) - 1u; break; }
                            }
                        } else {                                  // AC band, refinement
                            int k = Ss;
                            if (eobrun == 0) {
                                for (; k <= Se; ++k) {
                                    const int rs = br.decode(ac[ta[i]]);
                                    int r = rs >> 4, sz = rs & 15, val = 0;
                                    if (sz) val = br.bits(1) ? p1 : m1;              // (sz is 1: a new coefficient of magnitude 1 << Al)
                                    else if (r != 15) { eobrun = (1u << r) + (r ? br.bits(r) : 0u); break; }
                                    // skip r coefficients whose history is zero; the non-zero ones on the way take a correction bit each
                                    for (; k <= Se; ++k) {
                                        short& cf = blk[h_nat[k]];
                                        if (cf != 0) {
                                            if (br.bits(1) && (cf & p1) == 0) cf = (short)(cf + (cf >= 0 ? p1 : m1));
                                        } else if (--r < 0) break;
                                    }
                                    if (val && k <= 63) blk[h_nat[k]] = (short)val;
                                }
                            }
                            if (eobrun > 0) {                     // the rest of the band: correction bits only
                                for (; k <= Se; ++k) {
                                    short& cf = blk[h_nat[k]];
                                    if (cf != 0 && br.bits(1) && (cf & p1) == 0) cf = (short)(cf + (cf >= 0 ? p1 : m1));
                                }
                                --eobrun;
                            }
                        }
                    }
            }
        }
        if (br.starved) return fail("truncated scan (the file ends inside a progressive scan)");
        // the scan's end: the marker the reader ran into, or the next marker in the file
        if (br.marker >= 0) pos = br.pos - 2;
        else {
            size_t q = br.pos;
            while (q + 1 < n && !(f[q] == 0xFF && f[q + 1] != 0x00 && f[q + 1] != 0xFF)) ++q;
            pos = q;
        }
        while (pos + 1 < n && f[pos] == 0xFF && f[pos + 1] >= 0xD0 && f[pos + 1] <= 0xD7) pos += 2;      // (a stray restart marker at the end of a scan)
    }
    if (!saw_eoi) return fail("truncated file (no EOI marker)");
    if (scans == 0) return fail("no scan");
    return GDT_OK;
}

// the two host steps for a LIST of files on a few threads (they are independent per file and memory-bound; one call instead of 2n also spares a
// scripting host its per-call overhead)
template <typename F>
void for_each_file(int n, int threads, F&& fn) {
    threads = std::max(1, std::min(threads, std::min(n, 16)));
    if (threads == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int gdt_jpeg_parse(const unsigned char* file, size_t nbytes, gdt_jpeg_info* info) {
    GDT_REQUIRE(file != nullptr && info != nullptr, "jpeg: null argument");
    return parse_impl(file, nbytes, info);
}

int gdt_jpeg_extract_scan(const unsigned char* file, size_t nbytes, const gdt_jpeg_info* info, unsigned char* dst, unsigned int* seg_off) {
    GDT_REQUIRE(file != nullptr && info != nullptr && dst != nullptr && seg_off != nullptr, "jpeg: null argument");
    GDT_REQUIRE(info->scan_offset > 0 && info->scan_offset <= nbytes && info->nsegments >= 1, "jpeg: info was not filled by gdt_jpeg_parse");
    GDT_REQUIRE(!info->progressive, "jpeg: a progressive file has no single scan to extract (gdt_jpeg_progressive_coefficients)");
    const ScanWalk w = walk_scan(file, nbytes, info->scan_offset, dst, info->scan_capacity - 16, seg_off, info->nsegments - 1);
    GDT_REQUIRE(!w.overflow && w.restarts + 1 == info->nsegments && scan_capacity_for(w.data_bytes) == info->scan_capacity,
                "jpeg: the file does not match the info");
    seg_off[info->nsegments] = (unsigned int)w.data_bytes;
    memset(dst + w.data_bytes, 0, info->scan_capacity - w.data_bytes);
    return GDT_OK;
}

int gdt_jpeg_parse_batch(const unsigned char* const* files, const size_t* nbytes, int n, gdt_jpeg_info* infos, int* status, int threads) {
    GDT_REQUIRE(files != nullptr && nbytes != nullptr && infos != nullptr && status != nullptr && n >= 0, "jpeg: null argument");
    for_each_file(n, threads, [&](int i) { status[i] = files[i] ? parse_impl(files[i], nbytes[i], &infos[i]) : GDT_ERR_INVALID; });
    return GDT_OK;
}

int gdt_jpeg_extract_scan_batch(const unsigned char* const* files, const size_t* nbytes, const gdt_jpeg_info* infos, int n, unsigned char* dst,
                                const size_t* dst_off, unsigned int* seg_off, const size_t* seg_index, int threads) {
    GDT_REQUIRE(files != nullptr && nbytes != nullptr && infos != nullptr && dst != nullptr && dst_off != nullptr && seg_off != nullptr && seg_index != nullptr && n >= 0,
                "jpeg: null argument");
    std::vector<int> rc(n, GDT_OK);
    for_each_file(n, threads, [&](int i) { rc[i] = gdt_jpeg_extract_scan(files[i], nbytes[i], &infos[i], dst + dst_off[i], seg_off + seg_index[i]); });
    for (int i = 0; i < n; ++i)
        if (rc[i] != GDT_OK) { gdt_set_error("jpeg: file " + std::to_string(i) + " of the list does not match its info"); return rc[i]; }
    return GDT_OK;
}

int gdt_jpeg_decode_workspace_bytes(const gdt_jpeg_item* items, int n, size_t* bytes) {
    GDT_REQUIRE(bytes != nullptr, "bytes");
    Plan p;
    // (mode 0 has the larger footprint: its pieces are the shorter ones)
    const int rc = make_plan(items, n, 0, p);
    if (rc != GDT_OK) return rc;
    *bytes = p.total;
    return GDT_OK;
}

int gdt_jpeg_decode_u8_batch(const gdt_jpeg_item* items, int n, int mode, void* workspace, size_t workspace_bytes, void* stream) {
    Plan p;
    const int rc = make_plan(items, n, mode, p);
    if (rc != GDT_OK) return rc;
    GDT_REQUIRE(workspace != nullptr, "jpeg: null workspace");
    if (workspace_bytes < p.total) { gdt_set_error("jpeg: workspace too small"); return GDT_ERR_WORKSPACE; }
    char* ws = (char*)(((uintptr_t)workspace + ALIGN - 1) / ALIGN * ALIGN);
    return run_decode(items, n, mode, p, ws, (hipStream_t)stream);
}

int gdt_jpeg_progressive_coefficients(const unsigned char* file, size_t nbytes, const gdt_jpeg_info* info, short* coef) {
    GDT_REQUIRE(file != nullptr && info != nullptr && coef != nullptr, "jpeg: null argument");
    GDT_REQUIRE((info->ncomp == 1 || info->ncomp == 3 || info->ncomp == 4) && info->mcus_x > 0 && info->mcus_y > 0 && info->blocks_per_mcu >= 1, "jpeg: info was not filled by gdt_jpeg_parse");
    return progressive_impl(file, nbytes, *info, coef);
}

int gdt_jpeg_progressive_coefficients_batch(const unsigned char* const* files, const size_t* nbytes, const gdt_jpeg_info* infos, int n, short* coef,
                                            const size_t* coef_off, int* status, int threads) {
    GDT_REQUIRE(files != nullptr && nbytes != nullptr && infos != nullptr && coef != nullptr && coef_off != nullptr && status != nullptr && n >= 0, "jpeg: null argument");
    for_each_file(n, threads, [&](int i) { status[i] = files[i] ? gdt_jpeg_progressive_coefficients(files[i], nbytes[i], &infos[i], coef + coef_off[i]) : GDT_ERR_INVALID; });
    return GDT_OK;
}

static int plan_coef(const gdt_jpeg_info* infos, int n, const size_t* coef_off, unsigned char* const* dst, Plan& p) {
    GDT_REQUIRE(infos != nullptr && n >= 1 && n <= 65535, "jpeg: 1..65535 images per call");
    p.imgs.resize(n); p.tabs.resize(n);
    for (int i = 0; i < n; ++i) {
        const gdt_jpeg_info& f = infos[i];
        GDT_REQUIRE((f.ncomp == 1 || f.ncomp == 3 || f.ncomp == 4) && f.width > 0 && f.height > 0 && f.mcus_x > 0 && f.mcus_y > 0 && f.blocks_per_mcu >= 1,
                    "jpeg: info was not filled by gdt_jpeg_parse");
        DImg& d = p.imgs[i];
        memset(&d, 0, sizeof(d));
        d.width = f.width; d.height = f.height; d.ncomp = f.ncomp; d.B = f.blocks_per_mcu;
        d.mcus_x = f.mcus_x; d.mcus_y = f.mcus_y; d.hs0 = f.ncomp >= 3 ? f.hs[0] : 1; d.vs0 = f.ncomp >= 3 ? f.vs[0] : 1; d.ycck = f.ncomp == 4 && f.adobe_transform == 2;
        d.total_blocks = (unsigned)((long long)f.mcus_x * f.mcus_y * f.blocks_per_mcu);
        d.coef_off = coef_off ? coef_off[i] : 0;
        for (int c = 0; c < f.ncomp; ++c) {
            const int hs = c == 0 ? d.hs0 : 1, vs = c == 0 ? d.vs0 : 1;
            d.pitch[c] = f.mcus_x * hs * 8;
            d.dw[c] = (f.width * hs + d.hs0 - 1) / d.hs0;
            d.dh[c] = (f.height * vs + d.vs0 - 1) / d.vs0;
            d.plane_off[c] = p.plane_bytes;
            p.plane_bytes += ((size_t)d.pitch[c] * f.mcus_y * vs * 8 + 15) / 16 * 16;
            memcpy(p.tabs[i].quant[c], f.quant[f.tq[c]], 128);
        }
        d.dst = dst ? dst[i] : nullptr;
        p.max_blocks = std::max(p.max_blocks, d.total_blocks);
        p.max_pixels = std::max(p.max_pixels, (unsigned)f.width * (unsigned)f.height);
        GDT_REQUIRE((long long)f.width * f.height < (1LL << 31), "jpeg: image too large");
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + ALIGN - 1) / ALIGN * ALIGN; return at; };
    p.off_imgs = take(p.imgs.size() * sizeof(DImg));
    p.off_tabs = take(p.tabs.size() * sizeof(DTab));
    p.off_planes = take(p.plane_bytes);
    p.total = o + ALIGN;
    return GDT_OK;
}

int gdt_jpeg_decode_coef_workspace_bytes(const gdt_jpeg_info* infos, int n, size_t* bytes) {
    GDT_REQUIRE(bytes != nullptr, "bytes");
    Plan p;
    const int rc = plan_coef(infos, n, nullptr, nullptr, p);
    if (rc != GDT_OK) return rc;
    *bytes = p.total;
    return GDT_OK;
}

int gdt_jpeg_decode_coef_u8_batch(const gdt_jpeg_info* infos, const short* coef_dev, const size_t* coef_off, unsigned char* const* dst_hwc, int n,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    GDT_REQUIRE(coef_dev != nullptr && coef_off != nullptr && dst_hwc != nullptr && workspace != nullptr, "jpeg: null argument");
    Plan p;
    const int rc = plan_coef(infos, n, coef_off, dst_hwc, p);
    if (rc != GDT_OK) return rc;
    for (int i = 0; i < n; ++i) GDT_REQUIRE(dst_hwc[i] != nullptr, "jpeg: null output buffer");
    if (workspace_bytes < p.total) { gdt_set_error("jpeg: workspace too small"); return GDT_ERR_WORKSPACE; }
    char* ws = (char*)(((uintptr_t)workspace + ALIGN - 1) / ALIGN * ALIGN);
    hipStream_t st = (hipStream_t)stream;
    DImg* d_imgs = (DImg*)(ws + p.off_imgs);
    DTab* d_tabs = (DTab*)(ws + p.off_tabs);
    unsigned char* d_planes = (unsigned char*)(ws + p.off_planes);
    GDT_CHECK_HIP(hipMemcpyAsync(d_imgs, p.imgs.data(), p.imgs.size() * sizeof(DImg), hipMemcpyHostToDevice, st));
    GDT_CHECK_HIP(hipMemcpyAsync(d_tabs, p.tabs.data(), p.tabs.size() * sizeof(DTab), hipMemcpyHostToDevice, st));
    GDT_CHECK_HIP(hipStreamSynchronize(st));                    // (the staging vectors live in this frame only)
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((p.max_blocks + 63u) / 64u, n), dim3(64), 0, st, d_imgs, d_tabs, coef_dev, d_planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((p.max_pixels + 255u) / 256u, n), dim3(256), 0, st, d_imgs, d_planes);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // extern "C"
