/*
 * ort_parse.cpp -- scene ingestion on the host: .scn grammar, ASCII PLY, OBJ, mesh
 * placement, camera basis.  Own implementation; the behaviour it must reproduce is the
 * reference's (paths relative to /root/reference/code):
 *   number lexer      parser.cpp:158-250   (f64 digit accumulation scaled by (double)0.1f)
 *   .scn tokens       parser.cpp:985-1125  grammar parser.cpp:1184-1446
 *   PLY               parser.cpp:269-570
 *   OBJ               parser.cpp:574-982
 *   placement         macos_main.mm:382-413, math.h:745-793
 *   camera            macos_main.mm:550-556
 * Where the reference asserts (platform.h:16-20) this returns ORT_ERR_PARSE.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <float.h>

#include "ort_detmath.h"
#include "ort_scene.h"

namespace ort {

namespace {

/* ---- character cursor -------------------------------------------------------------- */
struct Cursor {
    const char *at;
    const char *end;

    bool done() const { return at >= end; }
    char peek(size_t k = 0) const { return (at + k < end) ? at[k] : '\0'; }
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\r'; } /* tabs are NOT space: parser.cpp:143-156 */
    void skip_space() { while (at < end && is_space(*at)) ++at; }
    void skip_word() { while (at < end && !is_space(*at)) ++at; }
    void skip_line() { while (at < end && *at != '\n' && *at != '\r') ++at; }
    /* the reference matches keywords by prefix (parser.cpp:13-32) */
    bool starts_with(const char *kw) const {
        const char *p = at;
        while (*kw) {
            if (p >= end || *p != *kw) return false;
            ++p; ++kw;
        }
        return true;
    }
};

/* A numeric literal as the reference reads it.  The integer and float views share
   storage there (a union), which matters for "integer mantissa + exponent" literals. */
struct Number {
    bool is_float = false;
    union { int32_t i; float f; } v;
};

Number lex_number(Cursor *c) {
    Number n;
    n.v.i = 0;
    double digits = 0;
    double scale = 10.0f;
    bool exponent = false;
    while (!c->done()) {
        char ch = *c->at;
        if (ch >= '0' && ch <= '9') {
            digits *= 10;
            digits += ch - '0';
        } else if (ch == '.') {
            n.is_float = true;
        } else if (ch == 'e') {
            exponent = true;
            break;
        } else {
            break;
        }
        if (n.is_float) scale *= 0.1f; /* f64 *= (double)0.1f, once per consumed character incl. the '.' */
        ++c->at;
    }
    if (n.is_float) n.v.f = (float)(digits * scale);
    else n.v.i = (int32_t)digits;
    if (exponent) {
        ++c->at; /* 'e' */
        bool plus = (c->peek() == '+');
        ++c->at; /* sign character, consumed unconditionally */
        float ev = 0;
        while (!c->done() && *c->at >= '0' && *c->at <= '9') {
            ev *= 10;
            ev += (*c->at - '0');
            ++c->at;
        }
        n.v.f *= ort_powf(10.0f, (plus ? 1 : -1) * ev);
    }
    return n;
}

/* ---- .scn -------------------------------------------------------------------------- */
enum class Tok { None, Screen, Camera, Ambient, Light, Sphere, Brdf, Box, Cylinder, Mesh, F32, I32, Str, B, Q, Z };

struct ScnToken {
    Tok type = Tok::None;
    const char *start = nullptr;
    int32_t len = 0;
    float f = 0;
    int32_t i = 0;
};

ScnToken next_scn_token(Cursor *c) {
    ScnToken t;
    c->skip_space();
    if (c->done()) return t;
    static const struct { const char *kw; Tok type; } keywords[] = {
        {"screen", Tok::Screen}, {"camera", Tok::Camera}, {"ambient", Tok::Ambient}, {"light", Tok::Light},
        {"sphere", Tok::Sphere}, {"brdf", Tok::Brdf},     {"box", Tok::Box},         {"cylinder", Tok::Cylinder},
        {"mesh", Tok::Mesh}};
    bool matched = false;
    for (const auto &k : keywords) {
        if (c->starts_with(k.kw)) { t.type = k.type; matched = true; break; }
    }
    char ch = c->peek();
    if (matched) {
    } else if (ch == 'b' && c->peek(1) == ' ') {
        t.type = Tok::B;
    } else if (ch == 'q' && c->peek(1) == ' ') {
        t.type = Tok::Q;
    } else if (ch == 'z' && c->peek(1) == ' ') {
        t.type = Tok::Z;
    } else if ((ch >= 'a' && ch < 'z') || (ch >= 'A' && ch < 'Z')) { /* sic: 'z'/'Z' excluded */
        t.type = Tok::Str;
        t.start = c->at;
        c->skip_word();
        t.len = (int32_t)(c->at - t.start);
    } else if (ch == '-' || (ch >= '0' && ch <= '9')) {
        bool neg = (ch == '-');
        if (neg) ++c->at; /* then read a number whatever follows */
        Number n = lex_number(c);
        if (n.is_float) {
            t.type = Tok::F32;
            t.f = n.v.f;
            if (neg) t.f *= -1.0f;
        } else {
            t.type = Tok::I32;
            t.i = n.v.i;
            if (neg) t.i *= -1;
        }
    }
    c->skip_word(); /* every token ends by running to the next whitespace */
    return t;
}

struct ScnParser {
    Cursor c;
    std::string *err;
    bool failed = false;

    ScnToken expect(Tok type, const char *what) {
        ScnToken t = next_scn_token(&c);
        if (t.type != type && !failed) {
            failed = true;
            *err = std::string("scn: expected ") + what;
        }
        return t;
    }
    float f32(const char *what) { return expect(Tok::F32, what).f; }
    int32_t i32(const char *what) { return expect(Tok::I32, what).i; }
};

/* ---- PLY --------------------------------------------------------------------------- */
enum class PlyTok { None, Element, Vertex, Face, EndHeader, Property, F32, I32 };
struct PlyToken { PlyTok type = PlyTok::None; float f = 0; int32_t i = 0; };

PlyToken next_ply_token(Cursor *c) {
    PlyToken t;
    c->skip_space();
    if (c->done()) return t;
    if (c->starts_with("element")) { t.type = PlyTok::Element; c->skip_word(); }
    else if (c->starts_with("vertex")) { t.type = PlyTok::Vertex; c->skip_word(); }
    else if (c->starts_with("face")) { t.type = PlyTok::Face; c->skip_word(); }
    else if (c->starts_with("end_header")) { t.type = PlyTok::EndHeader; c->skip_word(); }
    else if (c->starts_with("property")) { t.type = PlyTok::Property; c->skip_line(); }
    else {
        char ch = c->peek();
        if (ch == '-' || (ch >= '0' && ch <= '9')) {
            bool neg = (ch == '-');
            if (neg) ++c->at;
            Number n = lex_number(c);
            if (n.is_float) { t.type = PlyTok::F32; t.f = n.v.f; if (neg) t.f *= -1.0f; }
            else { t.type = PlyTok::I32; t.i = n.v.i; if (neg) t.i *= -1; }
        } else {
            c->skip_line(); /* unknown header line */
        }
    }
    return t;
}

float ply_value(const PlyToken &t) { return t.type == PlyTok::I32 ? (float)t.i : t.f; }

int load_ply(const std::vector<char> &file, HostMesh *mesh, std::string *err) {
    Cursor c{file.data(), file.data() + file.size()};
    uint32_t vertex_count = 0, property_lines = 0;
    bool header_done = false;
    while (!c.done() && !header_done) {
        PlyToken t = next_ply_token(&c);
        switch (t.type) {
        case PlyTok::Element: {
            PlyToken what = next_ply_token(&c);
            if (what.type == PlyTok::Vertex) {
                PlyToken n = next_ply_token(&c);
                if (n.type != PlyTok::I32) { *err = "ply: vertex count is not an integer"; return ORT_ERR_PARSE; }
                /* a vertex line is at least six bytes ("0 0 0\n"): a count the file cannot hold (or a negative one)
                   is a malformed file, not a 50 GB allocation */
                if (n.i < 0 || (size_t)n.i > file.size() / 6u) { *err = "ply: vertex count exceeds the file"; return ORT_ERR_PARSE; }
                vertex_count = (uint32_t)n.i;
            } else if (what.type != PlyTok::Face) {
                *err = "ply: unknown element";
                return ORT_ERR_PARSE;
            }
        } break;
        case PlyTok::Property: ++property_lines; break;
        case PlyTok::EndHeader: header_done = true; break;
        default: break;
        }
    }
    /* every property line counts, the face list's included, minus one (parser.cpp:420-433) */
    uint32_t per_vertex = property_lines - 1u;
    if (!header_done || property_lines == 0 || per_vertex < 3) { *err = "ply: malformed header"; return ORT_ERR_PARSE; }

    mesh->vertices.resize((size_t)vertex_count * 3);
    for (uint32_t v = 0; v < vertex_count; ++v) {
        PlyToken x = next_ply_token(&c), y = next_ply_token(&c), z = next_ply_token(&c);
        auto numeric = [](const PlyToken &t) { return t.type == PlyTok::F32 || t.type == PlyTok::I32; };
        if (!numeric(x) || !numeric(y) || !numeric(z)) { *err = "ply: vertex is not numeric"; return ORT_ERR_PARSE; }
        mesh->vertices[3 * (size_t)v + 0] = ply_value(x);
        mesh->vertices[3 * (size_t)v + 1] = ply_value(y);
        mesh->vertices[3 * (size_t)v + 2] = ply_value(z);
        c.skip_line(); /* remaining properties of this vertex */
    }
    /* faces: "k i0 i1 ... " fan-triangulated as (i0, previous, next) */
    for (;;) {
        if (c.done()) break;
        Cursor look = c;
        if (next_ply_token(&look).type == PlyTok::None) break;
        PlyToken k = next_ply_token(&c);
        if (k.type != PlyTok::I32 || k.i < 3) { *err = "ply: bad face vertex count"; return ORT_ERR_PARSE; }
        PlyToken a = next_ply_token(&c), b = next_ply_token(&c), d = next_ply_token(&c);
        if (a.type != PlyTok::I32 || b.type != PlyTok::I32 || d.type != PlyTok::I32) {
            *err = "ply: face index is not an integer";
            return ORT_ERR_PARSE;
        }
        mesh->indices.push_back((uint32_t)a.i);
        mesh->indices.push_back((uint32_t)b.i);
        mesh->indices.push_back((uint32_t)d.i);
        for (int32_t extra = 1; extra < k.i - 2; ++extra) {
            uint32_t prev = mesh->indices.back();
            PlyToken nx = next_ply_token(&c);
            mesh->indices.push_back((uint32_t)a.i);
            mesh->indices.push_back(prev);
            mesh->indices.push_back((uint32_t)nx.i);
        }
    }
    return ORT_OK;
}

/* ---- OBJ --------------------------------------------------------------------------- */
enum class ObjTok { None, V, Vn, Vt, F, I32, F32, Slash };
struct ObjToken { ObjTok type = ObjTok::None; float f = 0; int32_t i = 0; };

/* stuck = the reference's tokenizer neither recognised nor consumed the character
   (it would spin forever in its parse loop); reported as a parse error here. */
ObjToken next_obj_token(Cursor *c, bool *stuck) {
    ObjToken t;
    c->skip_space();
    if (c->done()) return t;
    bool neg = false;
    if (c->peek() == '-') { neg = true; ++c->at; }
    if (c->starts_with("v ")) { t.type = ObjTok::V; c->skip_word(); }
    else if (c->starts_with("vt ")) { t.type = ObjTok::Vt; c->skip_word(); }
    else if (c->starts_with("vn ")) { t.type = ObjTok::Vn; c->skip_word(); }
    else if (c->starts_with("f ")) { t.type = ObjTok::F; c->skip_word(); }
    else if (c->starts_with("mtllib ") || c->starts_with("o ") || c->starts_with("usemtl ") || c->starts_with("g ") ||
             c->starts_with("body") || c->peek() == '#') {
        c->skip_line();
    } else if (c->peek() == '/') {
        t.type = ObjTok::Slash;
        ++c->at;
    } else if (c->peek() >= '0' && c->peek() <= '9') {
        Number n = lex_number(c);
        if (n.is_float) { t.type = ObjTok::F32; t.f = n.v.f; if (neg) t.f *= -1.0f; }
        else { t.type = ObjTok::I32; t.i = n.v.i; if (neg) t.i *= -1; }
    } else if (!c->done() && !neg) {
        *stuck = true;
    }
    return t;
}

float obj_value(const ObjToken &t) { return t.type == ObjTok::I32 ? (float)t.i : t.f; }

int load_obj(const std::vector<char> &file, HostMesh *mesh, std::string *err) {
    bool stuck = false;
    /* pass 1: which of v / vt / vn appear decides the face layout globally (parser.cpp:753-771) */
    bool has_v = false, has_vt = false, has_vn = false;
    {
        Cursor c{file.data(), file.data() + file.size()};
        while (!c.done()) {
            ObjToken t = next_obj_token(&c, &stuck);
            if (stuck) { *err = "obj: unsupported line (the reference tokenizer would not advance)"; return ORT_ERR_PARSE; }
            if (t.type == ObjTok::V) has_v = true;
            else if (t.type == ObjTok::Vt) has_vt = true;
            else if (t.type == ObjTok::Vn) has_vn = true;
        }
    }
    if (!has_v) { *err = "obj: no positions"; return ORT_ERR_PARSE; }
    if (has_vt && !has_vn) { *err = "obj: v/vt faces are not implemented by the reference"; return ORT_ERR_UNSUPPORTED; }
    /* numbers per face corner: 1 (v), 2 (v//vn), 3 (v/vt/vn) */
    const int per_corner = has_vn ? (has_vt ? 3 : 2) : 1;

    Cursor c{file.data(), file.data() + file.size()};
    while (!c.done()) {
        ObjToken t = next_obj_token(&c, &stuck);
        if (t.type == ObjTok::V) {
            ObjToken x = next_obj_token(&c, &stuck), y = next_obj_token(&c, &stuck), z = next_obj_token(&c, &stuck);
            auto numeric = [](const ObjToken &k) { return k.type == ObjTok::F32 || k.type == ObjTok::I32; };
            if (!numeric(x) || !numeric(y) || !numeric(z)) { *err = "obj: position is not numeric"; return ORT_ERR_PARSE; }
            mesh->vertices.push_back(obj_value(x));
            mesh->vertices.push_back(obj_value(y));
            mesh->vertices.push_back(obj_value(z));
        } else if (t.type == ObjTok::F) {
            /* the first three corners are read as fixed token runs, then the fan */
            int32_t first[3];
            for (int corner = 0; corner < 3; ++corner) {
                ObjToken v = next_obj_token(&c, &stuck);
                first[corner] = v.i;
                /* v//vn: 2 slashes + 1 number; v/vt/vn: slash number slash number */
                int more = (per_corner == 1) ? 0 : (per_corner == 2 ? 3 : 4);
                for (int k = 0; k < more; ++k) next_obj_token(&c, &stuck);
            }
            mesh->indices.push_back((uint32_t)(first[0] - 1));
            mesh->indices.push_back((uint32_t)(first[1] - 1));
            mesh->indices.push_back((uint32_t)(first[2] - 1));
            uint32_t numbers_seen = 0;
            for (;;) {
                Cursor look = c;
                bool ignore = false;
                ObjToken nx = next_obj_token(&look, &ignore);
                if (nx.type == ObjTok::I32) {
                    if (numbers_seen % (uint32_t)per_corner == 0) {
                        uint32_t prev = mesh->indices.back();
                        mesh->indices.push_back((uint32_t)(first[0] - 1));
                        mesh->indices.push_back(prev);
                        mesh->indices.push_back((uint32_t)(nx.i - 1));
                    }
                    ++numbers_seen;
                    next_obj_token(&c, &stuck);
                } else if (nx.type == ObjTok::Slash && per_corner > 1) {
                    next_obj_token(&c, &stuck);
                } else {
                    break;
                }
            }
        }
        if (stuck) { *err = "obj: unsupported line"; return ORT_ERR_PARSE; }
    }
    return ORT_OK;
}

/* ---- placement (macos_main.mm:382-413) ----------------------------------------------- */
struct Rot3 { float m[3][3]; };

/* math.h:771-793: rotation matrix of quaternion (x,y,z,w), f32, this exact expression order */
Rot3 quat_matrix(float x, float y, float z, float w) {
    Rot3 r;
    r.m[0][0] = 1.0f - 2 * y * y - 2 * z * z; r.m[0][1] = 2 * x * y - 2 * w * z; r.m[0][2] = 2 * x * z + 2 * w * y;
    r.m[1][0] = 2 * x * y + 2 * w * z; r.m[1][1] = 1.0f - 2 * x * x - 2 * z * z; r.m[1][2] = 2 * y * z - 2 * w * x;
    r.m[2][0] = 2 * x * z - 2 * w * y; r.m[2][1] = 2 * y * z + 2 * w * x; r.m[2][2] = 1 - 2 * x * x - 2 * y * y;
    return r;
}
inline void rotate(const Rot3 &r, float v[3]) {
    float x = r.m[0][0] * v[0] + r.m[0][1] * v[1] + r.m[0][2] * v[2];
    float y = r.m[1][0] * v[0] + r.m[1][1] * v[1] + r.m[1][2] * v[2];
    float z = r.m[2][0] * v[0] + r.m[2][1] * v[1] + r.m[2][2] * v[2];
    v[0] = x; v[1] = y; v[2] = z;
}

void place_mesh(HostMesh *mesh, const float translate[3], float scale, float degree, const float quat_xyzw[4]) {
    /* the "z <deg>" token rotates about Y (macos_main.mm:399), angle = 0.0174533f * deg
       (math.h:745-769: q0 = cos(rad/2), q = axis * sin(rad/2)) */
    float rad = 0.0174533f * degree;
    float cs = ort_cosf(rad / 2), sn = ort_sinf(rad / 2);
    Rot3 about_y = quat_matrix(0.0f * sn, 1.0f * sn, 0.0f * sn, cs);
    Rot3 from_file = quat_matrix(quat_xyzw[0], quat_xyzw[1], quat_xyzw[2], quat_xyzw[3]);
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {FLT_MIN, FLT_MIN, FLT_MIN}; /* sic: smallest positive float, macos_main.mm:383 */
    size_t n = mesh->vertices.size() / 3;
    for (size_t i = 0; i < n; ++i) {
        float *v = &mesh->vertices[3 * i];
        v[0] *= scale; v[1] *= scale; v[2] *= scale;
        rotate(about_y, v);
        rotate(from_file, v);
        v[0] += translate[0]; v[1] += translate[1]; v[2] += translate[2];
        for (int k = 0; k < 3; ++k) {
            lo[k] = (lo[k] < v[k]) ? lo[k] : v[k];
            hi[k] = (hi[k] > v[k]) ? hi[k] : v[k];
        }
    }
    if (n) {
        mesh->aabb_min = {lo[0], lo[1], lo[2]};
        mesh->aabb_max = {hi[0], hi[1], hi[2]};
    }
}

/* The reference takes the text after the FIRST '.' of the whole path (get_extension,
   parser.cpp:91-108, used at macos_main.mm:351-366; its own note: "does not work if there was a
   directory with ."), compared by prefix.  Here: after the last '.' of the file name -- the same
   answer wherever the reference works, and a defined one for "../data/x.obj" or "/tmp/a.b/x.ply". */
bool extension_is(const std::string &path, const char *ext) {
    size_t slash = path.find_last_of('/');
    size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return false;
    const char *p = path.c_str() + dot + 1;
    while (*p && *ext) {
        if (*p != *ext) return false;
        ++p; ++ext;
    }
    return true;
}

} // namespace

int read_file(const char *path, std::vector<char> *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return ORT_ERR_IO;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(out->data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == out->size() ? ORT_OK : ORT_ERR_IO;
}

int parse_scn_text(const char *text, size_t size, const char *base_dir, Scene *s, std::string *err) {
    ScnParser p;
    p.c = Cursor{text, text + size};
    p.err = err;
    const size_t capacity = 100; /* parser.h:195-208 */
    s->materials.clear();
    s->materials.push_back(ort_material{}); /* material 0 = "no hit" (parser.cpp:1187) */

    while (!p.c.done() && !p.failed) {
        ScnToken t = next_scn_token(&p.c);
        switch (t.type) {
        case Tok::Screen:
            s->screen_width = p.i32("screen width (integer)");
            s->screen_height = p.i32("screen height (integer)");
            break;
        case Tok::Camera: {
            float x = p.f32("camera x"), y = p.f32("camera y"), z = p.f32("camera z");
            p.expect(Tok::B, "'b'");
            s->camera_height_ratio = p.f32("camera height ratio");
            p.expect(Tok::Q, "'q'");
            float qw = p.f32("camera q.w"), qx = p.f32("camera q.x"), qy = p.f32("camera q.y"), qz = p.f32("camera q.z");
            s->camera_p = {x, y, z};
            s->camera_quat[0] = qx; s->camera_quat[1] = qy; s->camera_quat[2] = qz; s->camera_quat[3] = qw;
        } break;
        case Tok::Ambient: {
            float r = p.f32("ambient r"), g = p.f32("ambient g"), b = p.f32("ambient b");
            s->ambient = {r, g, b};
        } break;
        case Tok::Light: {
            int32_t r = p.i32("light r (integer)"), g = p.i32("light g (integer)"), b = p.i32("light b (integer)");
            ort_material m{};
            m.is_light = 1;
            m.emit = {(float)r, (float)g, (float)b};
            s->materials.push_back(m);
        } break;
        case Tok::Brdf: {
            ort_material m{};
            float dr = p.f32("brdf diffuse r"), dg = p.f32("brdf diffuse g"), db = p.f32("brdf diffuse b");
            float sr = p.f32("brdf specular r"), sg = p.f32("brdf specular g"), sb = p.f32("brdf specular b");
            int32_t alpha = p.i32("brdf alpha (integer)");
            Cursor look = p.c;
            ScnToken peek = next_scn_token(&look);
            if (peek.type == Tok::F32 || peek.type == Tok::I32) {
                float tr = p.f32("brdf transmission r"), tg = p.f32("brdf transmission g"), tb = p.f32("brdf transmission b");
                m.transmission = {tr, tg, tb};
                m.ior = p.f32("brdf ior");
            }
            m.diffuse = {dr, dg, db};
            m.specular[0] = sr; m.specular[1] = sg; m.specular[2] = sb; m.specular[3] = (float)alpha;
            s->materials.push_back(m);
        } break;
        case Tok::Sphere: {
            ort_sphere sp{};
            float x = p.f32("sphere x"), y = p.f32("sphere y"), z = p.f32("sphere z");
            sp.center = {x, y, z};
            sp.r = p.f32("sphere r");
            sp.mat = (uint32_t)s->materials.size() - 1u;
            s->spheres.push_back(sp);
            if (s->materials[sp.mat].is_light) s->lights.push_back(ort_light{1u, (uint32_t)s->spheres.size() - 1u});
        } break;
        case Tok::Box: {
            ort_box b{};
            float x = p.f32("box x"), y = p.f32("box y"), z = p.f32("box z");
            float dx = p.f32("box dx"), dy = p.f32("box dy"), dz = p.f32("box dz");
            b.min = {x, y, z};
            b.max = {x + dx, y + dy, z + dz};
            b.mat = (uint32_t)s->materials.size() - 1u;
            s->boxes.push_back(b);
        } break;
        case Tok::Cylinder: {
            ort_cylinder cy{};
            float bx = p.f32("cylinder base x"), by = p.f32("cylinder base y"), bz = p.f32("cylinder base z");
            float ax = p.f32("cylinder axis x"), ay = p.f32("cylinder axis y"), az = p.f32("cylinder axis z");
            cy.base = {bx, by, bz};
            cy.axis = {ax, ay, az};
            cy.r = p.f32("cylinder r");
            cy.mat = (uint32_t)s->materials.size() - 1u;
            s->cylinders.push_back(cy);
            /* every cylinder with a material goes on the light list, light or not (parser.cpp:1345-1348) */
            if (cy.mat) s->lights.push_back(ort_light{2u, (uint32_t)s->cylinders.size() - 1u});
        } break;
        case Tok::Mesh: {
            ScnToken name = p.expect(Tok::Str, "mesh file name");
            float translate[3];
            translate[0] = p.f32("mesh tx"); translate[1] = p.f32("mesh ty"); translate[2] = p.f32("mesh tz");
            float scale = p.f32("mesh scale");
            float degree = 0;
            ScnToken sel = next_scn_token(&p.c);
            if (sel.type == Tok::Z) {
                ScnToken a = next_scn_token(&p.c);
                if (a.type == Tok::F32) degree = a.f;
                else if (a.type == Tok::I32) degree = (float)a.i;
                p.expect(Tok::Q, "'q'");
            } else if (sel.type != Tok::Q && !p.failed) {
                p.failed = true;
                *err = "scn: mesh orientation must start with 'z' or 'q'";
            }
            float quat[4] = {0, 0, 0, 0}; /* xyzw; file order is w x y z */
            const int order[4] = {3, 0, 1, 2};
            for (int k = 0; k < 4; ++k) {
                ScnToken q = next_scn_token(&p.c);
                if (q.type == Tok::F32) quat[order[k]] = q.f;
                else if (q.type == Tok::I32) quat[order[k]] = (float)q.i;
            }
            if (p.failed) break;
            std::string path = std::string(base_dir ? base_dir : "") + std::string(name.start, (size_t)name.len);
            HostMesh mesh;
            mesh.mat = (uint32_t)s->materials.size() - 1u;
            std::vector<char> file;
            if (read_file(path.c_str(), &file) != ORT_OK) { *err = "cannot read mesh file " + path; return ORT_ERR_IO; }
            int rc;
            if (extension_is(path, "ply")) rc = load_ply(file, &mesh, err);
            else if (extension_is(path, "obj")) rc = load_obj(file, &mesh, err);
            else { *err = "mesh file is neither .ply nor .obj: " + path; return ORT_ERR_UNSUPPORTED; }
            if (rc != ORT_OK) return rc;
            size_t nv = mesh.vertices.size() / 3;
            for (uint32_t ix : mesh.indices)
                if (ix >= nv) { *err = "mesh index out of range in " + path; return ORT_ERR_PARSE; }
            place_mesh(&mesh, translate, scale, degree, quat);
            s->meshes.push_back(std::move(mesh));
        } break;
        default: break; /* stray numbers / unknown words are skipped (parser.cpp:1115-1121) */
        }
        if (s->materials.size() > capacity || s->spheres.size() > capacity || s->boxes.size() > capacity ||
            s->cylinders.size() > capacity || s->meshes.size() >= capacity) {
            *err = "scn: more than 100 entries of one kind (parser.h:195-208)";
            return ORT_ERR_PARSE;
        }
    }
    return p.failed ? ORT_ERR_PARSE : ORT_OK;
}

void camera_basis(const Scene &s, int32_t width, int32_t height, ort_camera *out) {
    float rx = s.camera_height_ratio * ((float)width / height);
    Rot3 r = quat_matrix(s.camera_quat[0], s.camera_quat[1], s.camera_quat[2], s.camera_quat[3]);
    float ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
    rotate(r, ex); rotate(r, ey); rotate(r, ez);
    out->p = s.camera_p;
    out->x_axis = {rx * ex[0], rx * ex[1], rx * ex[2]};
    out->y_axis = {s.camera_height_ratio * ey[0], s.camera_height_ratio * ey[1], s.camera_height_ratio * ey[2]};
    out->z_axis = {ez[0], ez[1], ez[2]};
}

} // namespace ort
