// Layout of the constant tables the kernels index (built on the host by tables.cpp, uploaded once per context).
#pragma once
#include <stdint.h>

// Constant tables, device resident, built once per context (tables.cpp).
struct DevTables {
	int16_t dct[4][32 * 32];        // [log2N-2] N x N row-major HEVC core transform
	int16_t dst4[16];               // DST-VII 4x4
	int16_t dct_t[4][32 * 32];      // the same bases transposed (rows = basis columns), for the inverse stages
	int16_t dst4_t[16];
	int16_t dct_eo[4][32 * 32];     // inverse stages by even / odd input index (k_tu_chain): output k < N/2 has N/4 words (M[4i][k], M[4i+2][k]) then N/4 words (M[4i+1][k], M[4i+3][k])
	uint32_t scan[4][6][32 * 32];   // [scan_mode][log2N] coefficient scan order (mode 0 unused)
	int32_t quant[4][6][6][32 * 32];   // [log2N-2][list][qp%6]
	int32_t dequant[4][6][6][32 * 32];
	uint8_t blk2cg[4][6][64];       // [scan_mode][log2N][4x4 block in raster order] -> index of its coefficient group in scan order
	// The transform bases as MFMA operand fragments (enc_prims.h, tr_forward / tr_inverse on the matrix cores): binary16 bit patterns (every entry is an integer of
	// at most 90 in magnitude, exact in binary16), in the order the lanes of a wavefront hold them.
	//   frag16[dir][b][lane * 4 + e] = B[lane % 16][4 * (lane / 16) + e] of the basis b (0: DCT 4, 1: DCT 8, 2: DCT 16, 3: DST 4) padded with zeros to 16 x 16,
	//   dir 0: B = M (forward), dir 1: B = M transposed (inverse).
	//   fragp[dir][b][lane * 4 + e]: TWO blocks of size 4 (b = 0) or 8 (b = 1) side by side in one 16 x 16 tile, block h in rows / columns 8 h .. 8 h + 7 (block diagonal):
	//   B[lane % 8][4 * (lane / 16 % 2) + e] where lane % 16 / 8 == lane / 32, else 0 - the two halves of a wavefront transform a block each (PairGrp).
	uint16_t frag16[2][4][64 * 4];
	uint16_t fragp[2][2][64 * 4];
	//   frag32t[dir][R][K][lane * 4 + e] = B[16 R + lane % 16][16 K + 4 * (lane / 16) + e] of the 32 x 32 DCT: its four 16 x 16 quarters as fragments of the 16 x 16 x 16 product
	//   (the 32 x 32 transforms run as quarter tiles: eight accumulator registers in flight; a 32 x 32 x 8 chain held 64)
	uint16_t frag32t[2][2][2][64 * 4];
	//   fragq[dir][lane * 4 + e]: SIXTEEN 4 x 4 blocks in one 16 x 16 tile (block diagonal, DCT 4): B[lane % 4][e] where lane % 16 / 4 == lane / 16, else 0 -
	//   the merge evaluation of an 8 x 8 CU transforms the chroma blocks of all its candidates in one product (enc_quad.h); four 8 x 8 blocks use fragp[dir][1]
	uint16_t fragq[2][64 * 4];
};

