/* Drop-in include path for the reference's nntoolkitcore/signal/dft.h:
 * everything on the hot path is declared in nntoolkitcore_hip.h. */
#pragma once
#include "../../nntoolkitcore_hip.h"
