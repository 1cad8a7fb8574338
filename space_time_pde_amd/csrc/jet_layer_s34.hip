// k_layer instantiations for the stream configuration S1=3, S2=4 (round 5: BASELINE configs[4]'s equations name four second
// derivatives -- xx, yy, xy, tt -- and used to run padded to the (3,6) set; reference src/pde.py:137-142 evaluates exactly the
// derivatives the strings name).  Own translation unit: parallel compile.
#include "jet_layer_impl.h"
STPDE_DEFINE_LAYER_TU(3, 4)
