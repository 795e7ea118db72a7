// Include redirect for the batched build (see constraint_builder_2d.h next to this file).
#include "batched_constraint_builder_3d.h"
