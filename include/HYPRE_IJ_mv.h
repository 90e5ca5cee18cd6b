/* HYPRE_IJ_mv.h -- part of the hypre API subset; everything is declared in HYPRE.h */
#include "HYPRE.h"
