// k_layer instantiations for the stream configuration S1=0, S2=0 (own translation unit: parallel compile).
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(0, 0)
