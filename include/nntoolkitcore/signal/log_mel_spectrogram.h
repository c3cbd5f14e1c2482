/* Drop-in include path for the reference's nntoolkitcore/signal/log_mel_spectrogram.h:
 * everything on the hot path is declared in nntoolkitcore_hip.h. */
#pragma once
#include "../../nntoolkitcore_hip.h"
