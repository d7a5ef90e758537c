/* ORACLE -- TEST INFRASTRUCTURE ONLY (see modal_oracle.h): synthesis half and contact model. */
#ifndef MODAL_ORACLE_SYNTH_H
#define MODAL_ORACLE_SYNTH_H
#include "modal_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ModalEvent, src/audio/ModalAudio.h:28-37.  kind: 0 = Impact, 1 = Silence. */
typedef struct {
    uint32_t kind, object, ex_pos;
    float jx, jy, jz;
    float pulse_step, pulse_gamma, accel_amp;
    float click_b0, click_a1, click_a2;
} mo_event;

typedef struct mo_bank mo_bank;
mo_bank *mo_bank_create(float sample_rate, int use_double);
void mo_bank_free(mo_bank *);
/* shapes: [position][mode][xyz]; positions: [position][xyz]; indices: triangles over positions */
uint32_t mo_bank_add_object(mo_bank *, uint32_t entity, uint32_t n_modes, uint32_t n_pos, const float *shapes, const float *positions, uint32_t n_indices, const uint32_t *indices);
void mo_bank_tune_object(mo_bank *, int live, uint32_t object, uint32_t n, const float *freqs, const float *t60s, float radius_scale);
int mo_bank_set_shapes(mo_bank *, int live, uint32_t object, uint32_t n_modes, uint32_t n_pos, const float *shapes);
void mo_bank_set_gains(mo_bank *, int live, uint32_t object, float out_gain, float listener_gain);
void mo_bank_install(mo_bank *);
void mo_bank_set_renderers(mo_bank *, uint32_t count);
void mo_bank_set_click_gain(mo_bank *, float);
void mo_bank_set_max_impacts(mo_bank *, uint32_t);
int mo_bank_enqueue(mo_bank *, const mo_event *);
void mo_bank_render_f32(mo_bank *, float *out, uint32_t frames);
void mo_bank_render_f64(mo_bank *, double *out, uint32_t frames);
uint32_t mo_bank_num_objects(const mo_bank *);
uint32_t mo_bank_num_modes(const mo_bank *);
uint32_t mo_bank_active_impacts(const mo_bank *);
double mo_bank_modal_energy(const mo_bank *);
uint64_t mo_bank_events_dropped(const mo_bank *);
/* which: 0 CoeffRe 1 CoeffIm 2 StateRe 3 StateIm 4 RadiationGain 5 RadiationArea 6 DeflectionGain 7 OutPhaseIm
 * 8 OutPhaseRe 9 QuadCompliance 10 QuadDriveScale 11-13 ShapeX/Y/Z 14 OutGain 15 ListenerGain 16 RadiantRadius 17 DeflectionScale */
uint32_t mo_bank_column(const mo_bank *, int live, int which, double *out);
void mo_bank_object_state(const mo_bank *, uint32_t *tuned, uint32_t *live, uint8_t *ringing);

void mo_recoil_object_filter(double radius, double volume, double sample_rate, float out6[6]);
void mo_recoil_click_filter(double radius, double volume, double mass, double sample_rate, float out3[3]);

double mo_striker_mass(double density, float tip_radius, float length);
double mo_contact_patch_radius(double normal_force, double inv_effective_modulus, double combined_curvature);
double mo_static_penetration(double normal_force, double stiffness);
double mo_saturation_penetration(double combined_curvature, double nominal_area);
double mo_punch_stiffness(double inv_effective_modulus, double nominal_area);
void mo_inverse_inertia_tensor(const float inertia_diag[3], const float quat_wxyz[4], float out9_col_major[9]);
double mo_reduced_contact_mass(double mass, const float inv_inertia9[9], const float arm[3], const float dir[3], double impactor_inv_mass);
double mo_estimate_contact_time(double mass, const float inv_inertia9[9], const float arm[3], const float dir[3], double contact_speed,
                                const mo_material *object_material, double object_curvature, double nominal_area,
                                const mo_material *impactor_material, double impactor_curvature, double impactor_inv_mass,
                                double scale_ratio, double combined_roughness);
#ifdef __cplusplus
}
#endif
#endif
