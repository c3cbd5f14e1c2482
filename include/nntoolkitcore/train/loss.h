/* reference include path nntoolkitcore/train/loss.h: forwards to the single HIP drop-in header */
#ifndef NNTK_FWD_TRAIN_loss_H
#define NNTK_FWD_TRAIN_loss_H
#include "../../nntoolkitcore_hip.h"
#endif
