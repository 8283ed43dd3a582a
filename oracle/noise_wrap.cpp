// CPU ORACLE support (test infrastructure, NOT product code): C entry points over the
// reference's vendored libnoise 1.0.0 (+ its "bestest" quality patch), with the same
// parameters the reference's wrappers set (builtins/libnoise.cpp:34-88).  Built by
// oracle/build_ref.sh into oracle/_ref/libmmnoise.so from the sources where they lie.
#include "noise.h"

using namespace noise;

extern "C" float libnoise_perlin(int num_octaves, float persistence, float lacunarity, float x, float y, float z) {
    module::Perlin p;
    p.SetNoiseQuality(QUALITY_BESTEST);
    p.SetOctaveCount(num_octaves);
    p.SetLacunarity(lacunarity);
    p.SetPersistence(persistence);
    return p.GetValue(x, y, z);
}

extern "C" float libnoise_billow(int num_octaves, float persistence, float lacunarity, float x, float y, float z) {
    module::Billow p;
    p.SetNoiseQuality(QUALITY_BESTEST);
    p.SetOctaveCount(num_octaves);
    p.SetLacunarity(lacunarity);
    p.SetPersistence(persistence);
    return p.GetValue(x, y, z);
}

extern "C" float libnoise_ridged_multi(int num_octaves, float lacunarity, float x, float y, float z) {
    module::RidgedMulti p;
    p.SetNoiseQuality(QUALITY_BESTEST);
    p.SetOctaveCount(num_octaves);
    p.SetLacunarity(lacunarity);
    return p.GetValue(x, y, z);
}

extern "C" float libnoise_voronoi(float displacement, float x, float y, float z) {
    module::Voronoi p;
    p.SetDisplacement(displacement);
    return p.GetValue(x, y, z);
}
