// k_layer instantiations for the stream configuration S1=3, S2=1 (combined second-order stream).
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(3, 1)
