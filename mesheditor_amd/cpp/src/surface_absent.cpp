// The surface-contact model is optional in the reference (SURFACE_AUDIO off by default) and out of scope here: the
// hooks the bank would call are the no-ops of src/audio/SurfaceContactAbsent.cpp, so nothing is linked for them --
// the host mirror's RenderModal takes the impact-only kernel for every object.
