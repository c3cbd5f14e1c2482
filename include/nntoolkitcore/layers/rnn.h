/* Drop-in include path for the reference's nntoolkitcore/layers/rnn.h:
 * everything on the hot path is declared in nntoolkitcore_hip.h. */
#pragma once
#include "../../nntoolkitcore_hip.h"
