// k_layer instantiations for the stream configuration S1=3, S2=2 (own translation unit: parallel compile).
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(3, 2)
