// Scene file loading: the JSON scene format of the reference (scene.rs:64-619), .obj meshes as the
// wavefront_obj 7.0.0 crate presents them to scene.rs:501-561 (per-object 0-based indices, polygons
// as triangle fans), and image textures through decode_image_file.
#include <chrono>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "scene_host.hpp"

namespace rrt_host {
namespace {

// ---- a small JSON document model ----------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // file order; later duplicates win on lookup like serde_json

    const Json* get(const std::string& key) const {
        const Json* r = nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) r = &kv.second;
        return r;
    }
};

struct JsonParser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit JsonParser(const std::string& text) : s(text) {}

    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
    }
    bool fail(const std::string& m) {
        if (err.empty()) {
            size_t line = 1, col = 1;
            for (size_t k = 0; k < i && k < s.size(); ++k) {
                if (s[k] == '\n') { ++line; col = 1; } else ++col;
            }
            err = m + " at line " + std::to_string(line) + " column " + std::to_string(col);
        }
        return false;
    }
    bool parse_string(std::string* out) {
        if (i >= s.size() || s[i] != '"') return fail("expected string");
        ++i;
        out->clear();
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\') {
                if (i >= s.size()) return fail("EOF while parsing a string");
                char e = s[i++];
                switch (e) {
                    case '"': out->push_back('"'); break;
                    case '\\': out->push_back('\\'); break;
                    case '/': out->push_back('/'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'n': out->push_back('\n'); break;
                    case 'r': out->push_back('\r'); break;
                    case 't': out->push_back('\t'); break;
                    case 'u': {
                        if (i + 4 > s.size()) return fail("invalid escape");
                        unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
                        i += 4;
                        if (cp < 0x80) out->push_back((char)cp);
                        else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    default: return fail("invalid escape");
                }
            } else {
                out->push_back(c);
            }
        }
        if (i >= s.size()) return fail("EOF while parsing a string");
        ++i;
        return true;
    }
    bool parse_value(Json* out, int depth = 0) {
        if (depth > 128) return fail("recursion limit exceeded");
        ws();
        if (i >= s.size()) return fail("EOF while parsing a value");
        char c = s[i];
        if (c == '{') {
            out->kind = Json::Object;
            ++i;
            ws();
            if (i < s.size() && s[i] == '}') { ++i; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(&key)) return false;
                ws();
                if (i >= s.size() || s[i] != ':') return fail("expected `:`");
                ++i;
                Json v;
                if (!parse_value(&v, depth + 1)) return false;
                out->obj.emplace_back(std::move(key), std::move(v));
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return true; }
                return fail("expected `,` or `}`");
            }
        }
        if (c == '[') {
            out->kind = Json::Array;
            ++i;
            ws();
            if (i < s.size() && s[i] == ']') { ++i; return true; }
            for (;;) {
                Json v;
                if (!parse_value(&v, depth + 1)) return false;
                out->arr.push_back(std::move(v));
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return true; }
                return fail("expected `,` or `]`");
            }
        }
        if (c == '"') {
            out->kind = Json::String;
            return parse_string(&out->str);
        }
        if (s.compare(i, 4, "true") == 0) { out->kind = Json::Bool; out->b = true; i += 4; return true; }
        if (s.compare(i, 5, "false") == 0) { out->kind = Json::Bool; out->b = false; i += 5; return true; }
        if (s.compare(i, 4, "null") == 0) { out->kind = Json::Null; i += 4; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* start = s.c_str() + i;
            char* end = nullptr;
            out->num = std::strtod(start, &end);
            if (end == start) return fail("invalid number");
            i += (size_t)(end - start);
            out->kind = Json::Number;
            return true;
        }
        return fail("expected value");
    }
};

struct LoadError {
    std::string msg;
};
[[noreturn]] void bail(const std::string& m) { throw LoadError{m}; }

const Json& required(const Json& dict, const char* key) {  // scene.rs:162-175
    const Json* v = dict.kind == Json::Object ? dict.get(key) : nullptr;
    if (!v) bail(std::string("Required key ") + key + " is missing.");
    return *v;
}
float as_f32(const Json& j, const char* what) {
    if (j.kind != Json::Number) bail(std::string("invalid type: expected f32 for ") + what);
    return (float)j.num;
}
uint32_t as_u32(const Json& j, const char* what) {
    if (j.kind != Json::Number || j.num < 0 || j.num != std::floor(j.num) || j.num > 4294967295.0)
        bail(std::string("invalid type: expected u32 for ") + what);
    return (uint32_t)j.num;
}
const std::string& as_str(const Json& j, const char* what) {
    if (j.kind != Json::String) bail(std::string("invalid type: expected a string for ") + what);
    return j.str;
}
// Vector3 / Point3 (try_from Vec<f32>, vector.rs:146-156) and RGB (a struct: sequence or {x,y,z} map)
Vec3 as_vec3(const Json& j, const char* type_name) {
    if (j.kind == Json::Array) {
        if (j.arr.size() != 3) bail(std::string("Deserializing in to ") + type_name + " requires a Vec of length 3!");
        return mk(as_f32(j.arr[0], type_name), as_f32(j.arr[1], type_name), as_f32(j.arr[2], type_name));
    }
    if (j.kind == Json::Object && std::string(type_name) == "RGB")
        return mk(as_f32(required(j, "x"), "x"), as_f32(required(j, "y"), "y"), as_f32(required(j, "z"), "z"));
    bail(std::string("invalid type: expected a sequence for ") + type_name);
}
const std::string& type_of(const Json& dict) {  // scene.rs:177-187
    const Json& t = required(dict, "type");
    if (t.kind != Json::String) bail("Expected 'type' key to be a string");
    return t.str;
}

// ---- .obj ----------------------------------------------------------------------------------
struct ObjObject {
    std::string name;
    std::vector<double> v;   // xyz
    std::vector<double> vt;  // uv
    std::vector<int32_t> vidx, tidx;
};
std::vector<ObjObject> parse_obj(const std::string& text, const std::string& file) {
    std::vector<ObjObject> objs;
    size_t v_off = 0, t_off = 0;
    auto start = [&](const std::string& name) {
        if (!objs.empty()) {
            v_off += objs.back().v.size() / 3;
            t_off += objs.back().vt.size() / 2;
        }
        objs.emplace_back();
        objs.back().name = name;
    };
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "o") {
            std::string name;
            ls >> name;
            start(name);
        } else if (tag == "v") {
            if (objs.empty()) start("");
            double x, y, z;
            if (!(ls >> x >> y >> z)) bail("Error parsing vertex in " + file);
            objs.back().v.insert(objs.back().v.end(), {x, y, z});
        } else if (tag == "vt") {
            if (objs.empty()) start("");
            double u = 0, w = 0;
            if (!(ls >> u)) bail("Error parsing texture vertex in " + file);
            ls >> w;
            objs.back().vt.insert(objs.back().vt.end(), {u, w});
        } else if (tag == "f") {
            if (objs.empty()) start("");
            ObjObject& o = objs.back();
            std::vector<int32_t> vs, ts;
            std::string tok;
            while (ls >> tok) {
                long vi = 0, ti = 0;
                bool has_t = false;
                size_t s1 = tok.find('/');
                vi = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string tpart = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!tpart.empty()) {
                        ti = std::strtol(tpart.c_str(), nullptr, 10);
                        has_t = true;
                    }
                }
                long nv = (long)(o.v.size() / 3), nt = (long)(o.vt.size() / 2);
                vs.push_back((int32_t)(vi > 0 ? vi - 1 - (long)v_off : nv + vi));
                ts.push_back(has_t ? (int32_t)(ti > 0 ? ti - 1 - (long)t_off : nt + ti) : -1);
            }
            if (vs.size() < 3) bail("Only triangles are allowed in meshes, \n\t\t\t\t\t\t\tbut file " + file + ", object " + o.name + " had another type of primitive.");
            for (size_t k = 1; k + 1 < vs.size(); ++k) {
                o.vidx.insert(o.vidx.end(), {vs[0], vs[k], vs[k + 1]});
                o.tidx.insert(o.tidx.end(), {ts[0], ts[k], ts[k + 1]});
            }
        } else if (tag == "l" || tag == "p") {
            if (objs.empty()) start("");
            bail("Only triangles are allowed in meshes, \n\t\t\t\t\t\t\tbut file " + file + ", object " + objs.back().name + " had another type of primitive.");
        }
        // vn, g, s, usemtl, mtllib do not change what the scene loader consumes
    }
    return objs;
}

std::string read_file(const std::string& path, bool* ok) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        *ok = false;
        return "";
    }
    std::stringstream ss;
    ss << f.rdbuf();
    *ok = true;
    return ss.str();
}
std::string dir_of(const std::string& path) {
    size_t p = path.find_last_of('/');
    if (p == std::string::npos) return ".";
    if (p == 0) return "/";
    return path.substr(0, p);
}
std::string join(const std::string& dir, const std::string& rel) {
    if (!rel.empty() && rel[0] == '/') return rel;
    return dir + "/" + rel;
}

struct Loader {
    SceneBuild* sb;
    std::string base;
    std::map<std::string, int32_t> textures, materials;
    std::map<std::string, int32_t> image_cache;  // resolved path -> first texture using it (resources.rs:16-51)

    void check(const std::string& e) {
        if (!e.empty()) bail(e);
    }
    void transform(const Json& j, Vec3* tr, Vec3* rot, Vec3* scl) {  // transform.rs:14-31
        *tr = mk(0, 0, 0);
        *rot = mk(0, 0, 0);
        *scl = mk(1, 1, 1);
        const Json* t = j.get("transform");
        if (!t) return;
        if (const Json* v = t->get("translate")) *tr = as_vec3(*v, "Vector3");
        if (const Json* v = t->get("rotate")) *rot = as_vec3(*v, "Vector3");
        if (const Json* v = t->get("scale")) *scl = as_vec3(*v, "Vector3");
    }
    int32_t texture(const Json& j) {  // scene.rs:189-261
        if (j.kind != Json::Object) bail("Expected JSON object for value in Texture map");
        const std::string& ty = type_of(j);
        rrt_texture t{};
        int32_t id = -1;
        if (ty == "Constant") {
            Vec3 c = as_vec3(required(j, "color"), "RGB");
            t.type = RRT_TEX_CONSTANT; t.a = c.x; t.b = c.y; t.c = c.z;
            check(sb->add_texture(t, nullptr, &id));
        } else if (ty == "Test") {
            t.type = RRT_TEX_TEST;
            check(sb->add_texture(t, nullptr, &id));
        } else if (ty == "Checker") {
            float repeat = as_f32(required(j, "repeat"), "repeat");
            const Json& oddj = required(j, "odd");
            const Json& evenj = required(j, "even");
            int32_t odd = texture(oddj);
            int32_t even = texture(evenj);
            t.type = RRT_TEX_CHECKER; t.a = repeat; t.u0 = (uint32_t)odd; t.u1 = (uint32_t)even;
            check(sb->add_texture(t, nullptr, &id));
        } else if (ty == "Image") {
            std::string p = join(base, as_str(required(j, "image_path"), "image_path"));
            char* real = realpath(p.c_str(), nullptr);
            if (!real) bail("There was a problem finding the given image path: No such file or directory (os error 2)");
            std::string key(real);
            free(real);
            auto it = image_cache.find(key);
            if (it != image_cache.end()) {
                t = sb->textures[(size_t)it->second];  // same decoded image shared
                sb->textures.push_back(t);
                id = (int32_t)sb->textures.size() - 1;
            } else {
                uint32_t w, h, ch;
                std::vector<uint8_t> px;
                std::string e = decode_image_file(key, &w, &h, &ch, &px);
                if (!e.empty()) bail("Could not open image: " + e);
                t.type = RRT_TEX_IMAGE; t.u0 = w; t.u1 = h; t.u2 = ch;
                check(sb->add_texture(t, px.data(), &id));
                image_cache[key] = id;
            }
        } else if (ty == "Noise") {
            t.type = RRT_TEX_NOISE; t.a = as_f32(required(j, "scale"), "scale");
            check(sb->add_texture(t, nullptr, &id));
        } else if (ty == "Turbulence") {
            t.type = RRT_TEX_TURBULENCE;
            t.a = as_f32(required(j, "scale"), "scale");
            t.u0 = as_u32(required(j, "depth"), "depth");
            t.b = as_f32(required(j, "omega"), "omega");
            check(sb->add_texture(t, nullptr, &id));
        } else {
            bail("Unsupported texture type: " + ty);
        }
        return id;
    }
    int32_t tex_ref(const std::string& name, const char* prefix, const char* owner) {
        auto it = textures.find(name);
        if (it == textures.end()) bail(std::string("Missing ") + prefix + "Texture " + name + " for " + owner + ".");
        return it->second;
    }
    int32_t material(const Json& j) {  // scene.rs:263-407
        if (j.kind != Json::Object) bail("Expected JSON object for value in Materials map");
        const std::string& ty = type_of(j);
        int32_t id = -1;
        auto optional_bump = [&](const char* owner) -> int32_t {
            const Json* b = j.get("bump_map");
            if (!b || b->kind == Json::Null) return -1;
            return tex_ref(as_str(*b, "bump_map"), "bump map ", owner);
        };
        if (ty == "Lambert") {
            int32_t alb = tex_ref(as_str(required(j, "albedo"), "albedo"), "", "Lambert");
            check(sb->add_material(RRT_MAT_LAMBERT, alb, optional_bump("Lambert"), 0.0f, &id));
        } else if (ty == "Metal") {
            int32_t alb = tex_ref(as_str(required(j, "albedo"), "albedo"), "", "Metal");
            float rough = as_f32(required(j, "roughness"), "roughness");
            check(sb->add_material(RRT_MAT_METAL, alb, optional_bump("Metal"), rough, &id));
        } else if (ty == "Dielectric") {
            check(sb->add_material(RRT_MAT_DIELECTRIC, -1, -1, as_f32(required(j, "refractive_index"), "refractive_index"), &id));
        } else if (ty == "DiffuseLight") {
            check(sb->add_material(RRT_MAT_DIFFUSE_LIGHT, tex_ref(as_str(required(j, "emission"), "emission"), "", "DiffuseLight"), -1, 0.0f, &id));
        } else if (ty == "Isotropic") {
            check(sb->add_material(RRT_MAT_ISOTROPIC, tex_ref(as_str(required(j, "albedo"), "albedo"), "", "Isotropic"), -1, 0.0f, &id));
        } else {
            bail("Unsupported material type: " + ty);
        }
        return id;
    }
    int32_t mat_ref(const Json& j, const char* key, const char* what, const char* owner) {
        const std::string& name = as_str(required(j, key), key);
        auto it = materials.find(name);
        if (it == materials.end()) bail(std::string("Missing ") + what + " " + name + " for " + owner + ".");
        return it->second;
    }
    // scene.rs:409-437 ; as_boundary: the shape becomes a medium boundary instead of a scene shape.
    // Returns the number of shapes produced and, for boundaries, what kind the single one was.
    size_t shape(const Json& j, bool as_boundary, float density, int32_t phase) {
        if (j.kind != Json::Object) bail("Expected JSON object for value in Shapes array");
        const std::string& ty = type_of(j);
        if (ty == "Sphere") {  // scene.rs:449-470
            float radius = as_f32(required(j, "radius"), "radius");
            int32_t m = mat_ref(j, "material", "Material", "Sphere");
            Vec3 tr, rot, scl;
            transform(j, &tr, &rot, &scl);
            if (as_boundary) check(sb->add_medium_sphere(tr, rot, scl, radius, m, density, phase));
            else check(sb->add_sphere(tr, rot, scl, radius, m));
            return 1;
        }
        if (ty == "Mesh") {  // scene.rs:483-563
            const std::string& file = as_str(required(j, "file_path"), "file_path");
            const Json& cullj = required(j, "enable_backface_culling");
            if (cullj.kind != Json::Bool) bail("invalid type: expected a boolean for enable_backface_culling");
            int32_t m = mat_ref(j, "material", "Material", "Mesh");
            Vec3 tr, rot, scl;
            transform(j, &tr, &rot, &scl);
            bool ok;
            std::string text = read_file(join(base, file), &ok);
            if (!ok) bail("No such file or directory (os error 2)");
            std::vector<ObjObject> objs = parse_obj(text, file);
            size_t count = 0;
            for (const ObjObject& o : objs) count += o.vidx.size() / 3;
            if (as_boundary && count != 1) return count;
            for (const ObjObject& o : objs) {
                std::string e = sb->add_mesh(tr, rot, scl, (uint32_t)(o.v.size() / 3), o.v.data(), (uint32_t)(o.vt.size() / 2), o.vt.data(),
                                             (uint32_t)(o.vidx.size() / 3), o.vidx.data(), o.tidx.data(), cullj.b, m, as_boundary);
                if (!e.empty()) bail("Error creating Triangle for file " + file + ", object " + o.name + ": " + e);
            }
            if (as_boundary) check(sb->add_medium_last_boundary_triangle(density, phase));
            return count;
        }
        if (ty == "ConstantMedium") {  // scene.rs:573-602
            const Json& boundary = required(j, "boundary");
            float dens = as_f32(required(j, "density"), "density");
            const std::string& pname = as_str(required(j, "phase_func"), "phase_func");
            auto it = materials.find(pname);
            if (it == materials.end()) bail("Missing Phase Function " + pname + " for ConstantMedium.");
            if (shape(boundary, true, dens, it->second) != 1) bail("Only single shapes are allowed for constant mediums at the moment.");
            // this medium is itself a boundary: the one just added becomes the boundary of the one that asked
            if (as_boundary) check(sb->wrap_last_medium(density, phase));
            return 1;
        }
        bail("Unknown Shape 'type' " + ty + " given.");
    }
};

}  // namespace

std::string load_scene_file(const std::string& path, uint32_t ow, uint32_t oh, uint32_t os, float oaspect, int tree, SceneBuild* out) {
    const auto started = std::chrono::steady_clock::now();
    bool ok;
    std::string text = read_file(path, &ok);
    if (!ok) return "Failed to read scene spec file.";
    Json top;
    JsonParser p(text);
    if (!p.parse_value(&top)) return p.err;
    p.ws();
    if (p.i != text.size()) return "trailing characters";
    try {
        if (top.kind != Json::Object) bail("Top level scene spec is not a JSON object.");
        Loader L{out, dir_of(path), {}, {}, {}};
        const Json& lg = required(top, "Logistics");  // scene.rs:28-34
        out->res_x = ow ? ow : as_u32(required(lg, "resolution_x"), "resolution_x");
        out->res_y = oh ? oh : as_u32(required(lg, "resolution_y"), "resolution_y");
        out->samples = os ? os : as_u32(required(lg, "samples"), "samples");
        out->use_importance = true;
        if (const Json* u = lg.get("use_importance_sampling")) {
            if (u->kind == Json::Bool) out->use_importance = u->b;
            else if (u->kind != Json::Null) bail("invalid type: expected a boolean for use_importance_sampling");
        }
        const Json& cam = required(top, "Camera");  // camera.rs:66-88
        float aspect = as_f32(required(cam, "aspect_ratio"), "aspect_ratio");
        if (oaspect == oaspect && oaspect > 0.0f) aspect = oaspect;
        out->set_camera(as_vec3(required(cam, "position"), "Point3"), as_vec3(required(cam, "look_at"), "Point3"),
                        as_vec3(required(cam, "up"), "Vector3"), as_f32(required(cam, "fov"), "fov"), aspect,
                        as_f32(required(cam, "aperture"), "aperture"), as_f32(required(cam, "focus_distance"), "focus_distance"));
        const Json& tex = required(top, "Textures");
        if (tex.kind != Json::Object) bail("'Textures' is not a JSON object.");
        // serde_json's Map iterates in key order; a repeated key keeps its last value
        std::map<std::string, const Json*> by_name;
        for (const auto& kv : tex.obj) by_name[kv.first] = &kv.second;
        for (const auto& kv : by_name) L.textures[kv.first] = L.texture(*kv.second);
        const Json& mats = required(top, "Materials");
        if (mats.kind != Json::Object) bail("'Materials' is not a JSON object.");
        by_name.clear();
        for (const auto& kv : mats.obj) by_name[kv.first] = &kv.second;
        for (const auto& kv : by_name) L.materials[kv.first] = L.material(*kv.second);
        const Json& shapes = required(top, "Shapes");
        if (shapes.kind != Json::Array) bail("'Shapes' is not a JSON array.");
        for (const Json& s : shapes.arr) L.shape(s, false, 0.0f, -1);
        const Json& agg = required(top, "Aggregate");
        if (agg.kind != Json::String) bail("'Aggregate' is not a string.");
        if (agg.str != "List" && agg.str != "BVH") bail("Unknown Aggregate 'type' " + agg.str + " given.");
        // `tree` says where and how a "BVH" is built; a "List" stays a list
        out->load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count();
        std::string e = out->finalize(agg.str == "BVH" ? tree : RRT_AGGREGATE_LIST);
        if (!e.empty()) bail(e);
    } catch (const LoadError& e) {
        return e.msg;
    }
    return "";
}

}  // namespace rrt_host
