// Feature-plane extraction for gfx950: replaces nn/feature.py:10-57 (generate_input_planes,
// sym = 0) + GoBoard.get_board_data (board/go_board.py:468-478).
//
// HBM-bound byte kernel: P bytes in, 6*P fp32 out per position (2 025 B at 9x9, 9 025 B at
// 19x19).  A workgroup takes NPOS consecutive positions: their planes are ONE contiguous run of
// NPOS*6*P floats.  Phase 1 (one thread per cell): read the uint8 cell through the board symmetry,
// swap colours, write the cell's six plane values into an LDS image of that run.  Phase 2: the
// image goes out as 16-byte non-temporal stores (1 KB per wave instruction, every lane busy, no index
// arithmetic in the store loop).  (A wave per position with 4-byte stores left lanes 82..127 idle:
// 44 % of the HBM peak; computing plane / point from the flat index in the store loop was VALU-bound.)
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S>
__global__ __launch_bounds__(256) void featurize_kernel(const uint8_t *__restrict__ cells,
                                                        const int8_t *__restrict__ to_move,
                                                        const int32_t *__restrict__ prev_move,
                                                        const int32_t *__restrict__ moves,
                                                        const int8_t *__restrict__ sym,
                                                        int batch, float *__restrict__ planes) {
    constexpr int P = S * S;
    constexpr int W = S + 2;
    constexpr int NPOS = S == 9 ? 16 : 4;              // positions per workgroup
    constexpr int RUN = NPOS * 6 * P;                  // floats written by a workgroup (multiple of 4)
    static_assert(RUN % 4 == 0 && (6 * P * NPOS * 4) % 16 == 0, "16-byte stores");
    __shared__ __attribute__((aligned(16))) float img[RUN];   // [position][plane][point]
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NPOS;
    const int nvalid = batch - b0 < NPOS ? batch - b0 : NPOS;

    for (int e = tid; e < nvalid * P; e += 256) {
        const int bl = e / P, p = e - bl * P, b = b0 + bl;
        const int color = to_move[b];
        const int prev = prev_move[b];
        const bool pass_plane = moves[b] > 1 && prev == 0;   // feature.py:39
        // previous move as an on-board index (or -1): pos = x + y*W with a one-cell border
        int prev_idx = -1;
        if (!pass_plane && prev > 0) {
            const int py = prev / W - 1, px = prev % W - 1;
            if (py >= 0 && py < S && px >= 0 && px < S) prev_idx = py * S + px;
        }
        // output point p = (y, x) reads the cell the symmetry maps it to (go_board.py:80-104)
        const int y = p / S, x = p - y * S, n = S - 1;
        int ry = y, rx = x;
        switch (sym ? sym[b] : 0) {
            case 1: rx = n - x; break;
            case 2: ry = n - y; break;
            case 3: ry = n - y; rx = n - x; break;
            case 4: ry = x; rx = y; break;
            case 5: ry = n - x; rx = y; break;
            case 6: ry = x; rx = n - y; break;
            case 7: ry = n - x; rx = n - y; break;
            default: break;
        }
        const int sp = ry * S + rx;
        int c = cells[(size_t)b * P + sp];
        if (color == 2 && c != 0) c = 3 - c;                 // feature.py:24-25
        float *o = img + bl * 6 * P + p;
        o[0] = c == 0 ? 1.f : 0.f;
        o[P] = c == 1 ? 1.f : 0.f;
        o[2 * P] = c == 2 ? 1.f : 0.f;
        o[3 * P] = sp == prev_idx ? 1.f : 0.f;
        o[4 * P] = pass_plane ? 1.f : 0.f;
        o[5 * P] = color == 2 ? -1.f : 1.f;                  // feature.py:50-52
    }
    __syncthreads();
    float *dst = planes + (size_t)b0 * 6 * P;
    const int nfloat = nvalid * 6 * P;
    for (int e0 = tid * 4; e0 < nfloat; e0 += 1024) {
        if (e0 + 4 <= nfloat) {
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(img + e0), reinterpret_cast<f32x4 *>(dst + e0));
        } else {
            for (int e = e0; e < nfloat; ++e) dst[e] = img[e];
        }
    }
}

}  // namespace

extern "C" int tg_featurize_sym_dev(int board_size, const uint8_t *cells_dev, const int8_t *to_move_dev,
                                    const int32_t *prev_move_dev, const int32_t *moves_dev,
                                    const int8_t *sym_dev, int batch, float *planes_dev, void *stream);

extern "C" int tg_featurize_dev(int board_size, const uint8_t *cells_dev, const int8_t *to_move_dev,
                                const int32_t *prev_move_dev, const int32_t *moves_dev, int batch,
                                float *planes_dev, void *stream) {
    return tg_featurize_sym_dev(board_size, cells_dev, to_move_dev, prev_move_dev, moves_dev, nullptr, batch,
                                planes_dev, stream);
}

extern "C" int tg_featurize_sym_dev(int board_size, const uint8_t *cells_dev, const int8_t *to_move_dev,
                                    const int32_t *prev_move_dev, const int32_t *moves_dev,
                                    const int8_t *sym_dev, int batch, float *planes_dev, void *stream) {
    if (!cells_dev || !to_move_dev || !prev_move_dev || !moves_dev || !planes_dev)
        return tg::fail(TG_ERR_ARG, "tg_featurize_dev: null argument");
    if (batch <= 0) return batch == 0 ? TG_OK : tg::fail(TG_ERR_ARG, "tg_featurize_dev: negative batch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int npos = board_size == 9 ? 16 : 4;            // positions per workgroup (featurize_kernel NPOS)
    const dim3 grid((batch + npos - 1) / npos), block(256);
    if (board_size == 9)
        hipLaunchKernelGGL(featurize_kernel<9>, grid, block, 0, st, cells_dev, to_move_dev,
                           prev_move_dev, moves_dev, sym_dev, batch, planes_dev);
    else if (board_size == 13)
        hipLaunchKernelGGL(featurize_kernel<13>, grid, block, 0, st, cells_dev, to_move_dev,
                           prev_move_dev, moves_dev, sym_dev, batch, planes_dev);
    else if (board_size == 19)
        hipLaunchKernelGGL(featurize_kernel<19>, grid, block, 0, st, cells_dev, to_move_dev,
                           prev_move_dev, moves_dev, sym_dev, batch, planes_dev);
    else
        return tg::fail(TG_ERR_ARG, "tg_featurize_dev: board size %d not built", board_size);
    TG_HIP(hipGetLastError());
    return TG_OK;
}
