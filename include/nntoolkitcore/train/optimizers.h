/* reference include path nntoolkitcore/train/optimizers.h: forwards to the single HIP drop-in header */
#ifndef NNTK_FWD_TRAIN_optimizers_H
#define NNTK_FWD_TRAIN_optimizers_H
#include "../../nntoolkitcore_hip.h"
#endif
