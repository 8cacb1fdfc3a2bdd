// Syntax check of the reference-side adapter against the reference's own interface headers
// (uengine/urenderer.h:10-16, uengine/uutils.h:11-20,37-98, uengine/uscene.h): a signature drift on either side
// makes this translation unit fail to compile. Nothing is linked or run.
#include "ubdpthiprenderer.h"

#include <cstddef>
#include <type_traits>

static_assert(std::is_base_of<URenderer, UBDPTHipRenderer>::value, "the adapter is a URenderer");
static_assert(!std::is_abstract<UBDPTHipRenderer>::value, "every pure virtual of URenderer is overridden");
static_assert(sizeof(URenderParameters) == 56 && sizeof(ubdpt_render_params) == 56, "uutils.h:11-20");
static_assert(offsetof(URenderParameters, img_res_x) == offsetof(ubdpt_render_params, img_res_x), "layout");
static_assert(offsetof(URenderParameters, img_res_y) == offsetof(ubdpt_render_params, img_res_y), "layout");
static_assert(offsetof(URenderParameters, pixel_subdiv) == offsetof(ubdpt_render_params, pixel_subdiv), "layout");
static_assert(offsetof(URenderParameters, lens_subdiv) == offsetof(ubdpt_render_params, lens_subdiv), "layout");
static_assert(offsetof(URenderParameters, min_depth) == offsetof(ubdpt_render_params, min_depth), "layout");
static_assert(offsetof(URenderParameters, focus_plane_distance) == offsetof(ubdpt_render_params, focus_plane_distance), "layout");
static_assert(offsetof(URenderParameters, lens_size) == offsetof(ubdpt_render_params, lens_size), "layout");
static_assert(sizeof(glm::dvec3) == 3 * sizeof(double), "the sum buffer is [y][x] of three doubles");

// the way UEngine holds and drives a renderer (uengine/uengine.cpp:48-53, :57, :164, :204-208)
bool drive(std::shared_ptr<UScene> scene, const URenderParameters& params)
{
    std::unique_ptr<URenderer> renderer = std::make_unique<UBDPTHipRenderer>();
    if(!renderer->initialize(params, scene)) return false;
    auto buffer = std::make_shared<UBuffer2D<glm::dvec3>>(params.img_res_x, params.img_res_y);
    std::function<void(double)> cb;
    bool ok = renderer->renderPass(buffer, 0, 8, cb);
    renderer->stop();
    return ok;
}
