// state.h (host mirror) -- the five LIO fields of class state (include/state.h:18-22); the camera
// members (state.h:24-36) belong to the vision stage and are out of scope.
#pragma once
#include "srl_la.h"

namespace srlivo {

class state {
public:
    srl::Quat rotation;
    srl::Vec3 translation = srl::Vec3::Zero();
    srl::Vec3 velocity = srl::Vec3::Zero();
    srl::Vec3 ba = srl::Vec3::Zero();
    srl::Vec3 bg = srl::Vec3::Zero();
    state() = default;
    state(const srl::Quat &rotation_, const srl::Vec3 &translation_, const srl::Vec3 &velocity_, const srl::Vec3 &ba_,
          const srl::Vec3 &bg_)
        : rotation(rotation_), translation(translation_), velocity(velocity_), ba(ba_), bg(bg_) {}
};

}  // namespace srlivo
