// k_layer instantiations for the VALUE-TILE configuration S1=0, S2=3: four consecutive row tiles (value stream only) share
// one pass over the weights -- the forward-only (inference) path (own translation unit: parallel compile).
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(0, 3)
