// uflatsceneexport.h — file a reference maintainer adds to uengine/: the additive scene-export hook.
// UScene exposes geometry only through opaque virtuals (uengine/uscene.h:16-18, uengine/uobject.h:14-18), so the
// scene class that owns the data (uapp/scene.h) also implements this interface; the HIP renderer discovers it with
// dynamic_cast in initialize().
#ifndef UFLATSCENEEXPORT_H
#define UFLATSCENEEXPORT_H

#include <ubdpt.h>

class UFlatSceneExport
{
public:
    virtual ~UFlatSceneExport() = default;
    // fills `out` in UScene::objects() order; the description must stay valid until initialize() returns
    virtual bool flatten(ubdpt_scene_desc& out) = 0;
};

#endif // UFLATSCENEEXPORT_H
