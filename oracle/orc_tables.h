/*
 * orc_tables.h -- constant tables of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference (arkanis/video-encoder) delegates every bit of
 * this arithmetic to an un-vendored, un-pinned libx264 behind one call
 * (av_encode.c:970); it ships no tests, fixtures or golden vectors (SURVEY.md
 * section 4, section 8c).  These tables restate the published ITU-T H.264
 * Constrained-Baseline algorithm (CAVLC tables 9-5, 9-7, 9-8, 9-9, 9-10, the
 * quantiser constants of 8.5.9/8.5.12, the me(v) map of table 9-4) from the
 * implementer's knowledge of the standard.  tests/test_oracle_tables.py checks
 * them structurally (prefix-freeness, Kraft sums, permutation properties).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (video-encoder_amd/) carries its own
 * independently written tables.
 */
#ifndef ORC_TABLES_H
#define ORC_TABLES_H
#include <stdint.h>

/* 4x4 frame zig-zag: scan index -> raster index (4*row + col) */
static const uint8_t orc_zigzag4x4[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};

/* luma4x4BlkIdx -> (x,y) in 4x4 units inside the macroblock (figure 6-10) */
static const uint8_t orc_blk_x[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3};
static const uint8_t orc_blk_y[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};

/* quantiser multipliers MF[qp%6][class], class 0: (0,0),(0,2),(2,0),(2,2);
 * class 1: (1,1),(1,3),(3,1),(3,3); class 2: the rest */
static const uint16_t orc_quant_mf[6][3] = {
    {13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554},
    {9362, 3647, 5825},  {8192, 3355, 5243},  {7282, 2893, 4559}};
/* dequantiser normAdjust4x4 V[qp%6][class] (8.5.9) */
static const uint8_t orc_dequant_v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16},
                                            {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
/* coefficient class of raster position r = 4*row+col */
static inline int orc_pos_class(int r) {
    int row = r >> 2, col = r & 3;
    if (!(row & 1) && !(col & 1)) return 0;
    if ((row & 1) && (col & 1)) return 1;
    return 2;
}

/* QPc as a function of qPI (table 8-15), chroma_qp_index_offset = 0 */
static const uint8_t orc_chroma_qp[52] = {
    0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
    18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
    34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

/* me(v): coded_block_pattern -> codeNum for Inter macroblocks, 4:2:0
 * (table 9-4, column "Inter").  Stored as codeNum -> cbp; inverted at init. */
/* Table 9-4, column "Intra_4x4, Intra_8x8" (chroma_format_idc 1 or 2): codeNum -> coded_block_pattern */
static const uint8_t orc_golomb_to_intra_cbp[48] = {47, 31, 15, 0,  23, 27, 29, 30, 7,  11, 13, 14, 39, 43, 45, 46, 16, 3,  5,  10, 12, 19, 21, 26,
                                                    28, 35, 37, 42, 44, 1,  2,  4,  8,  17, 18, 20, 24, 6,  9,  22, 25, 32, 33, 34, 36, 40, 38, 41};
static const uint8_t orc_golomb_to_inter_cbp[48] = {
    0,  16, 1,  2,  4,  8,  32, 3,  5,  10, 12, 15, 47, 7,  11, 13,
    14, 6,  9,  31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46,
    17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41};

/* coeff_token (table 9-5).  Index [table][4*TotalCoeff + TrailingOnes].
 * table 0: 0<=nC<2, 1: 2<=nC<4, 2: 4<=nC<8, 3: nC>=8 (6-bit FLC). */
static const uint8_t orc_coeff_token_len[4][4 * 17] = {
    {1,  0,  0,  0,  6,  2,  0,  0,  8,  6,  3,  0,  9,  8,  7,  5,  10,
     9,  8,  6,  11, 10, 9,  7,  13, 11, 10, 8,  13, 13, 11, 9,  13, 13,
     13, 10, 14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15,
     14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16},
    {2,  0,  0,  0,  6,  2,  0,  0,  6,  5,  3,  0,  7,  6,  6,  4,  8,
     6,  6,  4,  8,  7,  7,  5,  9,  8,  8,  6,  11, 9,  9,  6,  11, 11,
     11, 7,  12, 11, 11, 9,  12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13,
     12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14},
    {4,  0,  0,  0,  6,  4,  0,  0,  6,  5,  4,  0,  6,  5,  5,  4,  7,
     5,  5,  4,  7,  5,  5,  4,  7,  6,  6,  4,  7,  6,  6,  4,  8,  7,
     7,  5,  8,  8,  7,  6,  9,  8,  8,  7,  9,  9,  8,  8,  9,  9,  9,
     8,  10, 9,  9,  9,  10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10},
    {6, 0, 0, 0, 6, 6, 0, 0, 6, 6, 6, 0, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
     6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
     6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6}};
static const uint8_t orc_coeff_token_bits[4][4 * 17] = {
    {1,  0,  0,  0,  5,  1,  0,  0,  7,  4,  1,  0,  7,  6,  5,  3,  7,
     6,  5,  3,  7,  6,  5,  4,  15, 6,  5,  4,  11, 14, 5,  4,  8,  10,
     13, 4,  15, 14, 9,  4,  11, 10, 13, 12, 15, 14, 9,  12, 11, 10, 13,
     8,  15, 1,  9,  12, 11, 14, 13, 8,  7,  10, 9,  12, 4,  6,  5,  8},
    {3,  0,  0,  0,  11, 2,  0,  0,  7,  7,  3,  0,  7,  10, 9,  5,  7,
     6,  5,  4,  4,  6,  5,  6,  7,  6,  5,  8,  15, 6,  5,  4,  11, 14,
     13, 4,  15, 10, 9,  4,  11, 14, 13, 12, 8,  10, 9,  8,  15, 14, 13,
     12, 11, 10, 9,  12, 7,  11, 6,  8,  9,  8,  10, 1,  7,  6,  5,  4},
    {15, 0,  0,  0,  15, 14, 0,  0,  11, 15, 13, 0,  8,  12, 14, 12, 15,
     10, 11, 11, 11, 8,  9,  10, 9,  14, 13, 9,  8,  10, 9,  8,  15, 14,
     13, 13, 11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9,  12, 8,  10, 13,
     8,  13, 7,  9,  12, 9,  12, 11, 10, 5,  8,  7,  6,  1,  4,  3,  2},
    {3,  0,  0,  0,  0,  1,  0,  0,  4,  5,  6,  0,  8,  9,  10, 11, 12,
     13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
     30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46,
     47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63}};

/* coeff_token for ChromaDCLevel, nC = -1.  Index [4*TotalCoeff + TrailingOnes] */
static const uint8_t orc_chroma_dc_coeff_token_len[4 * 5] = {2, 0, 0, 0, 6, 1, 0, 0, 6, 6,
                                                             3, 0, 6, 7, 7, 6, 6, 8, 8, 7};
static const uint8_t orc_chroma_dc_coeff_token_bits[4 * 5] = {1, 0, 0, 0, 7, 1, 0, 0, 4, 6,
                                                              1, 0, 3, 3, 2, 5, 2, 3, 2, 0};

/* total_zeros for 4x4 blocks (tables 9-7, 9-8): [tzVlcIndex-1][total_zeros] */
static const uint8_t orc_total_zeros_len[15][16] = {
    {1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9},
    {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6},
    {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6},
    {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5},
    {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5},
    {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6},
    {6, 5, 3, 3, 3, 2, 3, 4, 3, 6},
    {6, 4, 5, 3, 2, 2, 3, 3, 6},
    {6, 6, 4, 2, 2, 3, 2, 5},
    {5, 5, 3, 2, 2, 2, 4},
    {4, 4, 3, 3, 1, 3},
    {4, 4, 2, 1, 3},
    {3, 3, 1, 2},
    {2, 2, 1},
    {1, 1}};
static const uint8_t orc_total_zeros_bits[15][16] = {
    {1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1},
    {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0},
    {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0},
    {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0},
    {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0},
    {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0},
    {1, 1, 5, 4, 3, 3, 2, 1, 1, 0},
    {1, 1, 1, 3, 3, 2, 2, 1, 0},
    {1, 0, 1, 3, 2, 1, 1, 1},
    {1, 0, 1, 3, 2, 1, 1},
    {0, 1, 1, 2, 1, 3},
    {0, 1, 1, 1, 1},
    {0, 1, 1, 1},
    {0, 1, 1},
    {0, 1}};
/* total_zeros for chroma DC 2x2 (table 9-9a): [tzVlcIndex-1][total_zeros] */
static const uint8_t orc_chroma_dc_total_zeros_len[3][4] = {{1, 2, 3, 3}, {1, 2, 2, 0}, {1, 1, 0, 0}};
static const uint8_t orc_chroma_dc_total_zeros_bits[3][4] = {{1, 1, 1, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}};
/* run_before (table 9-10): [min(zerosLeft,7)-1][run_before] */
static const uint8_t orc_run_len[7][16] = {{1, 1},
                                           {1, 2, 2},
                                           {2, 2, 2, 2},
                                           {2, 2, 2, 3, 3},
                                           {2, 2, 3, 3, 3, 3},
                                           {2, 3, 3, 3, 3, 3, 3},
                                           {3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
static const uint8_t orc_run_bits[7][16] = {{1, 0},
                                            {1, 1, 0},
                                            {3, 2, 1, 0},
                                            {3, 2, 1, 1, 0},
                                            {3, 2, 3, 2, 1, 0},
                                            {3, 0, 1, 3, 2, 5, 4},
                                            {7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1}};

#endif
