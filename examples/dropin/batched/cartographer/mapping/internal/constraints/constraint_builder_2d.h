// Include redirect for the batched build: whoever includes the reference's
// constraint_builder_2d.h (PoseGraph2D in cartographer, the test main here) gets the batched
// ConstraintBuilder2D with the same public interface.
#include "batched_constraint_builder_2d.h"
