// src/bsdfs/merl.cpp for Mitsuba 3 (README.md:1 of the reference: "Merl ... brdf pluggin for ... Mitsuba 3.0")
#include "measured_bsdf.hpp"

NAMESPACE_BEGIN(mitsuba)
MI_IMPLEMENT_CLASS_VARIANT(MerlBSDF, BSDF)
NAMESPACE_END(mitsuba)
MI_EXPORT_PLUGIN(MerlBSDF, "MERL measured BRDF (MI355X / libmerl_hip)")
