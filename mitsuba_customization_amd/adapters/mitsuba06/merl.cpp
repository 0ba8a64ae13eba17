// plugins/merl.so for Mitsuba 0.6 (README.md:1 of the reference: "Merl ... brdf pluggin for Mitsuba 0.6")
#include "measured_bsdf.hpp"

MTS_NAMESPACE_BEGIN
MTS_IMPLEMENT_CLASS_S(MerlBSDF, false, BSDF)
MTS_NAMESPACE_END
MTS_EXPORT_PLUGIN(MerlBSDF, "MERL measured BRDF (MI355X / libmerl_hip)")
