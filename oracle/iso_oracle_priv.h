/* Private to the .c files of oracle/ -- TEST INFRASTRUCTURE ONLY (see iso_oracle.h). */
#ifndef ISO_ORACLE_PRIV_H
#define ISO_ORACLE_PRIV_H

struct iso_volume {
    int nx, ny, nz;
    int org[3];               /* global index of stored voxel (0,0,0); non-zero only for a tile (multiple of 8) */
    int n1o[3];               /* global 128^3 node coordinate of node1[0][0][0] */
    float* data;              /* [z][y][x] */
    int bx, by, bz;           /* 8^3 leaf counts */
    unsigned char* leaf;      /* [bz][by][bx] : leaf node exists */
    int mx, my, mz;           /* 128^3 node counts */
    unsigned char* node1;     /* [mz][my][mx] */
    int any_leaf;
    int nleaf;
    int nbox_min[3], nbox_max[3];   /* node-level bbox, max already offset by +1 */
    int abox_min[3], abox_max[3];   /* active voxel bbox */
    float max_value;
    double s, sinv, t[3];     /* index->world: w = i*s + t  (ScaleTranslateMap, TP/openvdb/math/Maps.h:1279-1305) */
    unsigned char* touched;   /* per-leaf-slot scratch for N_bricks_touched (benign write races) */
};

#endif
