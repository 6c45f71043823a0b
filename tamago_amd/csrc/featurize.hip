// Feature-plane extraction for gfx950: replaces nn/feature.py:10-57 (generate_input_planes,
// sym = 0) + GoBoard.get_board_data (board/go_board.py:468-478).
//
// HBM-bound byte kernel: P bytes in, 6*P fp32 out per position (2 025 B at 9x9, 9 025 B at
// 19x19).  One wavefront per position; every plane is written as one coalesced run of
// fp32, the uint8 cells are read once and kept in registers.
#include "common.h"

namespace {

template <int S>
__global__ __launch_bounds__(256) void featurize_kernel(const uint8_t *__restrict__ cells,
                                                        const int8_t *__restrict__ to_move,
                                                        const int32_t *__restrict__ prev_move,
                                                        const int32_t *__restrict__ moves,
                                                        const int8_t *__restrict__ sym,
                                                        int batch, float *__restrict__ planes) {
    constexpr int P = S * S;
    constexpr int W = S + 2;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const int color = to_move[b];
    const int prev = prev_move[b];
    const bool pass_plane = moves[b] > 1 && prev == 0;       // feature.py:39
    // previous move as an on-board index (or -1): pos = x + y*W with a one-cell border
    int prev_idx = -1;
    if (!pass_plane && prev > 0) {
        const int py = prev / W - 1, px = prev % W - 1;
        if (py >= 0 && py < S && px >= 0 && px < S) prev_idx = py * S + px;
    }
    const float side = color == 2 ? -1.f : 1.f;              // feature.py:50-52
    const int sy_ = sym ? sym[b] : 0;                        // board symmetry 0..7 (go_board.py:80-104)
    const uint8_t *src = cells + (size_t)b * P;
    float *dst = planes + (size_t)b * 6 * P;
    for (int p = lane; p < P; p += 64) {
        // output point p = (y, x) reads the cell the symmetry maps it to
        const int y = p / S, x = p - y * S, n = S - 1;
        int ry = y, rx = x;
        switch (sy_) {
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
        int c = src[sp];
        if (color == 2 && c != 0) c = 3 - c;                 // feature.py:24-25
        dst[p] = c == 0 ? 1.f : 0.f;
        dst[P + p] = c == 1 ? 1.f : 0.f;
        dst[2 * P + p] = c == 2 ? 1.f : 0.f;
        dst[3 * P + p] = sp == prev_idx ? 1.f : 0.f;
        dst[4 * P + p] = pass_plane ? 1.f : 0.f;
        dst[5 * P + p] = side;
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
    const dim3 grid((batch + 3) / 4), block(256);
    if (board_size == 9)
        hipLaunchKernelGGL(featurize_kernel<9>, grid, block, 0, st, cells_dev, to_move_dev,
                           prev_move_dev, moves_dev, sym_dev, batch, planes_dev);
    else if (board_size == 19)
        hipLaunchKernelGGL(featurize_kernel<19>, grid, block, 0, st, cells_dev, to_move_dev,
                           prev_move_dev, moves_dev, sym_dev, batch, planes_dev);
    else
        return tg::fail(TG_ERR_ARG, "tg_featurize_dev: board size %d not built", board_size);
    TG_HIP(hipGetLastError());
    return TG_OK;
}
