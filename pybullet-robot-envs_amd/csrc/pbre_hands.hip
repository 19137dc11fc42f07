// pbre_hands.hip -- the iCub with hands (icub_model_with_hands.sdf; 60 simulated DoF after the legs are pruned): the wide
// lane-group engine (pbre_wide.hip / pbre_wide_impl.hpp) instantiated for Shape128 -- one env per wavefront, two virtual lanes
// per physical lane (lanes_device.hpp DevLanes128), persistent per-env motor records, fingertip contact forces.  A separate
// translation unit because this instantiation is by far the largest kernel of the library.
//
// Replaces, per env (reference file:line): iCubHandsEnv.reset / apply_action / open_hand / pre_grasp / grasp /
// check_contact_fingertips / check_collision (icub_env_with_hands.py:85-318, icub_env.py:260-361) and the
// p.stepSimulation calls of the demo that drives them (examples/helloworlds/helloworld_icub.py:43-125).
#include "pbre_wide_impl.hpp"

namespace pbre {

WideEngine* make_hands_engine() { return new WideImpl<Shape128, DevLanes128>(); }

}  // namespace pbre
