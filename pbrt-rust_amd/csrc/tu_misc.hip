// tu_misc.hip -- scene preparation, camera-ray generation, routing, film, light grid and the parity helper kernels.
#include "kern_misc.h"
