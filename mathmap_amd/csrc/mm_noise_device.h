// Coherent-noise builtins (noise, noiseBillow, noiseRidgedMulti, voronoiCells) as __device__
// functions.  Restates the algorithms of libnoise 1.0.0 as the reference vendors and patches
// it (libnoisesrc-1.0.0.zip + libnoise-bestest.diff; wrappers builtins/libnoise.cpp:34-88):
// gradient noise over a 256-entry unit-vector table with the 7th-order "bestest" S-curve,
// fractal sums of octaves (Perlin / Billow / RidgedMulti) and Voronoi cells over integer
// value noise.  Everything is double precision like the library; the float arguments are
// promoted and the double result is rounded once on return.  The vector table
// (mm_noise_vectors) is emitted in front of this text by the code generator.
MM_DEV double mm_noise_scurve7(double a) {
    double a2 = a * a, a4 = a2 * a2, a5 = a4 * a, a6 = a4 * a2, a7 = a5 * a2;
    return -20.0 * a7 + 70.0 * a6 - 84.0 * a5 + 35.0 * a4;
}
MM_DEV double mm_noise_lerp(double n0, double n1, double a) { return ((1.0 - a) * n0) + (a * n1); }
MM_DEV double mm_noise_int32range(double n) {
    if (n >= 1073741824.0) return (2.0 * fmod(n, 1073741824.0)) - 1073741824.0;
    if (n <= -1073741824.0) return (2.0 * fmod(n, 1073741824.0)) + 1073741824.0;
    return n;
}
MM_DEV double mm_noise_gradient(double fx, double fy, double fz, int ix, int iy, int iz, int seed) {
    int vi = (int)(1619u * (unsigned)ix + 31337u * (unsigned)iy + 6971u * (unsigned)iz + 1013u * (unsigned)seed);
    vi ^= (vi >> 8);
    vi &= 0xff;
    const double xg = mm_noise_vectors[(vi << 2)], yg = mm_noise_vectors[(vi << 2) + 1], zg = mm_noise_vectors[(vi << 2) + 2];
    const double xp = fx - (double)ix, yp = fy - (double)iy, zp = fz - (double)iz;
    return ((xg * xp) + (yg * yp) + (zg * zp)) * 2.12;
}
MM_DEV double mm_noise_coherent(double x, double y, double z, int seed) {
    const int x0 = (x > 0.0 ? (int)x : (int)x - 1), x1 = x0 + 1;
    const int y0 = (y > 0.0 ? (int)y : (int)y - 1), y1 = y0 + 1;
    const int z0 = (z > 0.0 ? (int)z : (int)z - 1), z1 = z0 + 1;
    const double xs = mm_noise_scurve7(x - (double)x0), ys = mm_noise_scurve7(y - (double)y0), zs = mm_noise_scurve7(z - (double)z0);
    double n0, n1, ix0, ix1, iy0, iy1;
    n0 = mm_noise_gradient(x, y, z, x0, y0, z0, seed);
    n1 = mm_noise_gradient(x, y, z, x1, y0, z0, seed);
    ix0 = mm_noise_lerp(n0, n1, xs);
    n0 = mm_noise_gradient(x, y, z, x0, y1, z0, seed);
    n1 = mm_noise_gradient(x, y, z, x1, y1, z0, seed);
    ix1 = mm_noise_lerp(n0, n1, xs);
    iy0 = mm_noise_lerp(ix0, ix1, ys);
    n0 = mm_noise_gradient(x, y, z, x0, y0, z1, seed);
    n1 = mm_noise_gradient(x, y, z, x1, y0, z1, seed);
    ix0 = mm_noise_lerp(n0, n1, xs);
    n0 = mm_noise_gradient(x, y, z, x0, y1, z1, seed);
    n1 = mm_noise_gradient(x, y, z, x1, y1, z1, seed);
    ix1 = mm_noise_lerp(n0, n1, xs);
    iy1 = mm_noise_lerp(ix0, ix1, ys);
    return mm_noise_lerp(iy0, iy1, zs);
}
// Perlin / Billow: frequency 1, seed 0 (module defaults), octaves/persistence/lacunarity from the call
MM_DEV float libnoise_perlin(int octaves, float persistence, float lacunarity, float fx, float fy, float fz) {
    double x = fx, y = fy, z = fz, value = 0.0, cur = 1.0;
    const double lac = lacunarity, pers = persistence;
    for (int o = 0; o < octaves; ++o) {
        const double s = mm_noise_coherent(mm_noise_int32range(x), mm_noise_int32range(y), mm_noise_int32range(z), o);
        value += s * cur;
        x *= lac; y *= lac; z *= lac;
        cur *= pers;
    }
    return (float)value;
}
MM_DEV float libnoise_billow(int octaves, float persistence, float lacunarity, float fx, float fy, float fz) {
    double x = fx, y = fy, z = fz, value = 0.0, cur = 1.0;
    const double lac = lacunarity, pers = persistence;
    for (int o = 0; o < octaves; ++o) {
        double s = mm_noise_coherent(mm_noise_int32range(x), mm_noise_int32range(y), mm_noise_int32range(z), o);
        s = 2.0 * fabs(s) - 1.0;
        value += s * cur;
        x *= lac; y *= lac; z *= lac;
        cur *= pers;
    }
    value += 0.5;
    return (float)value;
}
MM_DEV float libnoise_ridged_multi(int octaves, float lacunarity, float fx, float fy, float fz) {
    double x = fx, y = fy, z = fz, value = 0.0, weight = 1.0, frequency = 1.0;
    const double lac = lacunarity, offset = 1.0, gain = 2.0;
    for (int o = 0; o < octaves; ++o) {
        double s = mm_noise_coherent(mm_noise_int32range(x), mm_noise_int32range(y), mm_noise_int32range(z), o & 0x7fffffff);
        s = fabs(s);
        s = offset - s;
        s *= s;
        s *= weight;
        weight = s * gain;
        if (weight > 1.0) weight = 1.0;
        if (weight < 0.0) weight = 0.0;
        value += s * pow(frequency, -1.0);     // spectral weight frequency^-h, h = 1
        frequency *= lac;
        x *= lac; y *= lac; z *= lac;
    }
    return (float)((value * 1.25) - 1.0);
}
MM_DEV int mm_noise_intvalue(int x, int y, int z, int seed) {
    unsigned n = (1619u * (unsigned)x + 31337u * (unsigned)y + 6971u * (unsigned)z + 1013u * (unsigned)seed) & 0x7fffffffu;
    n = (n >> 13) ^ n;
    return (int)((n * (n * n * 60493u + 19990303u) + 1376312589u) & 0x7fffffffu);
}
MM_DEV double mm_noise_value(int x, int y, int z, int seed) { return 1.0 - ((double)mm_noise_intvalue(x, y, z, seed) / 1073741824.0); }
MM_DEV float libnoise_voronoi(float displacement, float fx, float fy, float fz) {
    const double x = fx, y = fy, z = fz;
    const int xi = (x > 0.0 ? (int)x : (int)x - 1), yi = (y > 0.0 ? (int)y : (int)y - 1), zi = (z > 0.0 ? (int)z : (int)z - 1);
    double mind = 2147483647.0, xc = 0, yc = 0, zc = 0;
    for (int zq = zi - 2; zq <= zi + 2; zq++)
        for (int yq = yi - 2; yq <= yi + 2; yq++)
            for (int xq = xi - 2; xq <= xi + 2; xq++) {
                const double xp = xq + mm_noise_value(xq, yq, zq, 0), yp = yq + mm_noise_value(xq, yq, zq, 1),
                             zp = zq + mm_noise_value(xq, yq, zq, 2);
                const double xd = xp - x, yd = yp - y, zd = zp - z;
                const double dist = xd * xd + yd * yd + zd * zd;
                if (dist < mind) { mind = dist; xc = xp; yc = yp; zc = zp; }
            }
    return (float)(0.0 + ((double)displacement * mm_noise_value((int)floor(xc), (int)floor(yc), (int)floor(zc), 0)));
}
