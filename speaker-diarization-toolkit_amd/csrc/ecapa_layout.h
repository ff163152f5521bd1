/* Index space of sdk_ecapa_desc.off[] (byte offsets into the packed device weight blob).
 * Mirrored by weights_pack.py (slot() there must stay in sync with these macros).
 *
 * bf16 tensors ("W" slots of conv layers) are stored [C_out][taps * C_in_padded], tap-major,
 * K contiguous; everything else is fp32.  scale/shift = eval-mode BatchNorm folded to an affine. */
#ifndef SDK_ECAPA_LAYOUT_H
#define SDK_ECAPA_LAYOUT_H

#define EL_W 0
#define EL_B 1
#define EL_SCALE 2
#define EL_SHIFT 3

#define EL_BLK0 0                                   /* 4 slots */
#define EL_BLOCK_BASE(i) (4 + ((i) - 1) * 40)       /* i = 1..n_blocks, 40 slots each */
#define EL_TDNN1 0                                  /* +W,B,SCALE,SHIFT */
#define EL_RES2NET(j) (4 + 4 * (j))                 /* j = 0..6 */
#define EL_TDNN2 32
#define EL_SE_W1T 36                                /* fp32 [C][Cse]   */
#define EL_SE_B1 37
#define EL_SE_W2T 38                                /* fp32 [Cse][C]   */
#define EL_SE_B2 39
#define EL_TAIL_BASE(nb) (4 + (nb) * 40)
#define EL_MFA 0                                    /* +W,B,SCALE,SHIFT */
#define EL_ASP_WH 4                                 /* bf16 [A][Cm]    */
#define EL_ASP_WMS_T 5                              /* fp32 [2Cm][A]   */
#define EL_ASP_B 6
#define EL_ASP_SCALE 7
#define EL_ASP_SHIFT 8
#define EL_ASP_W2 9                                 /* bf16 [Cm][A]    */
#define EL_ASP_B2 10
#define EL_ASPBN_SCALE 11
#define EL_ASPBN_SHIFT 12
#define EL_FC_WT 13                                 /* fp32 [2Cm][E]   */
#define EL_FC_B 14
#define EL_ASP_W2PACK 15                            /* OPTIONAL (-1 = absent): EL_ASP_W2 in MFMA fragment order for the per-segment ASP kernel:
                                                      [channel block Cm/32][ks 8][lane 64][8] bf16, lane l holds
                                                      W2[32 blk + (l & 31)][16 ks + 8 (l >> 5) + 0..7] */

/* OPTIONAL (-1 = absent): the Res2Net chain's 128x128 k3 conv weights once more, in MFMA fragment order, so that a wave's
 * fragment load is 1 KiB contiguous instead of 16 row pieces of 64 B:
 *   [tap 3][channel block wq 4][column tile h 2][k-step ks 4][lane 64][8] bf16,
 *   lane l holds W[out = 32 wq + 16 h + (l & 15)][k = 128 tap + 32 ks + 8 (l >> 4) + 0..7]   (tap-major K as in the W slot) */
#define EL_CHAINPACK(i, j) (200 + ((i) - 1) * 8 + (j))   /* block i = 1..n_blocks (<= 4), conv j = 0..6: slots 200..231, clear of every legal
                                                            tail (EL_TAIL_BASE(4) + 15 = 179; round 2 had them at 160.., inside the 4-block tail) */

#endif
