// 4-lane build of ransac_simd.inc (see the Makefile for the -m flags)
#define VW 4
#define NS simd4
#include "ransac_simd.inc"
