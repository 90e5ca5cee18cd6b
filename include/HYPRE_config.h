/* HYPRE_config.h -- part of the hypre API subset; everything is declared in HYPRE.h */
#include "HYPRE.h"
