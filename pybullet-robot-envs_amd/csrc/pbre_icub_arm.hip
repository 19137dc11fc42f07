// pbre_icub_arm.hip -- iCubEnv used alone (icub_model.sdf, 20 simulated DoF after the legs are pruned; pbre_config.robot_level = 1):
// the wide lane-group engine (pbre_wide.hip / pbre_wide_impl.hpp) instantiated for ShapeIA -- one env per half-wave like the task
// envs' Shape32, plus the persistent per-env motor record (target, gain, force scale, maxVelocity per DoF) that PyBullet's
// POSITION_CONTROL motors keep between commands.  Its own translation unit so that it compiles beside the others.
//
// Replaces, per env (reference file:line): iCubEnv.reset (icub_env.py:91-151: motors at the initial positions with gain 0.2,
// apply_action(home hand pose) + one step with IK control), iCubEnv.apply_action(action, max_vel) (icub_env.py:259-360: joint
// commands clipped to the limits, gain 0.5; hand-pose commands through the IK over every joint, blocked joints at their rest
// poses, gain 0.2; `maxVelocity`) and the p.stepSimulation calls of a script that drives them.
#include "pbre_wide_impl.hpp"

namespace pbre {

WideEngine* make_icub_arm_engine() { return new WideImpl<ShapeIA, DevLanes32>(); }

}  // namespace pbre
