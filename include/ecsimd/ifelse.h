// ecsimd/ifelse.h -- if_else lives with swap_if (reference ifelse.h:15-49).
#ifndef ECSIMD_IFELSE_H
#define ECSIMD_IFELSE_H
#include <ecsimd/swap.h>
#endif
