// scene_load.cpp — scene TOML + Wavefront OBJ loading, mesh packing.
//
// Restates Scene::load_toml / SceneDescriptor::build_scene (reference src/scene.rs:233-441),
// Mesh::load (src/mesh.rs:29-82) and PackedMeshes::pack_meshes (src/mesh.rs:91-113).
// Schema (src/scene.rs:264-322): [[material]] name,color,roughness,metallic,emission;
// [[object]] with exactly one of [object.Sphere] material,pos,radius / [object.Plane]
// material,pos,forward,right / [object.Mesh] material,path; [camera] pos,yaw,pitch,fov_y (degrees).
// The TOML reader is a subset sufficient for that schema (tables, arrays of tables, dotted table
// headers, strings, numbers, booleans, possibly multi-line arrays, comments); integers are
// accepted where floats are expected, as serde does (assets/scenes/default.toml:3-4).
// OBJ (wavefront_obj 11.0.0 semantics, un-vendored): `o` starts an object, indices become
// 0-based and object-relative, polygons are fanned (v0, vi, vi+1), vt/usemtl/mtllib/s/g ignored,
// points and lines dropped, normals mandatory (mesh.rs:60-64).
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rsrt_host.h"
#include "host_math.h"

namespace rsrt_host {
int build_bvh_from_arrays(const rsrt_sphere *, uint32_t, const rsrt_plane_desc *, uint32_t, const rsrt_vec3 *, uint32_t,
                          const rsrt_triangle *, uint32_t, std::vector<rsrt_primitive_info> &, std::vector<rsrt_bvh_node> &,
                          uint32_t &);
}

struct rsrt_scene {
    std::vector<rsrt_material> materials;
    std::vector<rsrt_sphere> spheres;
    std::vector<rsrt_plane_desc> plane_descs;
    std::vector<rsrt_plane> planes;
    std::vector<rsrt_vec3> vertices, normals;
    std::vector<rsrt_triangle> triangles;
    std::vector<rsrt_primitive_info> primitives;
    std::vector<rsrt_bvh_node> nodes;
    rsrt_camera_desc camera;
    uint32_t depth = 0;
};

namespace {

// ---------------------------------------------------------------- minimal TOML value tree
struct Value;
typedef std::map<std::string, Value> Table;
struct Value {
    enum Kind { Num, Str, Bool, Arr, Tab, TabArr } kind = Num;
    double num = 0;
    bool is_int = false;
    std::string str;
    bool b = false;
    std::vector<Value> arr;       // Arr
    std::shared_ptr<Table> tab;   // Tab
    std::vector<std::shared_ptr<Table>> tabs; // TabArr
};

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct TomlParser {
    const std::string &s;
    size_t i = 0;
    int line = 1;
    explicit TomlParser(const std::string &src) : s(src) {}

    [[noreturn]] void fail(const std::string &msg) { throw ParseError("TOML parse error at line " + std::to_string(line) + ": " + msg); }
    bool eof() const { return i >= s.size(); }
    void skip_ws_inline() { while (!eof() && (s[i] == ' ' || s[i] == '\t')) i++; }
    void skip_comment() { if (!eof() && s[i] == '#') while (!eof() && s[i] != '\n') i++; }
    void skip_ws_all()
    {
        for (;;) {
            skip_ws_inline();
            skip_comment();
            if (!eof() && (s[i] == '\n' || s[i] == '\r')) { if (s[i] == '\n') line++; i++; continue; }
            break;
        }
    }
    std::string parse_key()
    {
        skip_ws_inline();
        std::string k;
        if (!eof() && s[i] == '"') return parse_string();
        while (!eof() && (isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-')) k += s[i++];
        if (k.empty()) fail("expected a key");
        return k;
    }
    std::string parse_string()
    {
        char q = s[i++];
        std::string out;
        while (!eof() && s[i] != q) {
            if (s[i] == '\n') fail("newline in string");
            if (q == '"' && s[i] == '\\') {
                i++;
                if (eof()) break;
                switch (s[i]) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case '\\': out += '\\'; break;
                case '"': out += '"'; break;
                default: fail("unsupported escape");
                }
                i++;
            } else out += s[i++];
        }
        if (eof()) fail("unterminated string");
        i++;
        return out;
    }
    Value parse_value()
    {
        skip_ws_inline();
        if (eof()) fail("expected a value");
        Value v;
        char c = s[i];
        if (c == '"' || c == '\'') { v.kind = Value::Str; v.str = parse_string(); return v; }
        if (c == '[') {
            i++;
            v.kind = Value::Arr;
            for (;;) {
                skip_ws_all();
                if (eof()) fail("unterminated array");
                if (s[i] == ']') { i++; break; }
                v.arr.push_back(parse_value());
                skip_ws_all();
                if (!eof() && s[i] == ',') { i++; continue; }
                if (!eof() && s[i] == ']') { i++; break; }
                fail("expected ',' or ']' in array");
            }
            return v;
        }
        if (s.compare(i, 4, "true") == 0) { i += 4; v.kind = Value::Bool; v.b = true; return v; }
        if (s.compare(i, 5, "false") == 0) { i += 5; v.kind = Value::Bool; v.b = false; return v; }
        size_t j = i;
        std::string tok;
        while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '+' || s[j] == '-' || s[j] == '.' || s[j] == '_')) {
            if (s[j] != '_') tok += s[j];
            j++;
        }
        if (tok.empty()) fail(std::string("unexpected character '") + c + "'");
        char *end = nullptr;
        errno = 0;
        double d = strtod(tok.c_str(), &end);
        if (*end != '\0') fail("invalid number '" + tok + "'");
        v.kind = Value::Num;
        v.num = d;
        v.is_int = tok.find_first_of(".eEn") == std::string::npos;
        i = j;
        return v;
    }
    Table parse()
    {
        Table root;
        Table *cur = &root;
        for (;;) {
            skip_ws_all();
            if (eof()) break;
            if (s[i] == '[') {
                bool arr = (i + 1 < s.size() && s[i + 1] == '[');
                i += arr ? 2 : 1;
                std::vector<std::string> path;
                for (;;) {
                    path.push_back(parse_key());
                    skip_ws_inline();
                    if (!eof() && s[i] == '.') { i++; continue; }
                    break;
                }
                if (eof() || s[i] != ']') fail("expected ']'");
                i++;
                if (arr) { if (eof() || s[i] != ']') fail("expected ']]'"); i++; }
                Table *t = &root;
                for (size_t k = 0; k < path.size(); k++) {
                    bool last = (k + 1 == path.size());
                    Value &slot = (*t)[path[k]];
                    if (last && arr) {
                        if (slot.kind != Value::TabArr) { if (slot.tab || !slot.arr.empty() || !slot.str.empty()) fail("key '" + path[k] + "' redefined"); slot.kind = Value::TabArr; }
                        slot.tabs.push_back(std::make_shared<Table>());
                        t = slot.tabs.back().get();
                    } else if (slot.kind == Value::TabArr) {
                        if (slot.tabs.empty()) fail("empty table array");
                        t = slot.tabs.back().get(); // [a.b] after [[a]] extends the last element
                    } else {
                        if (slot.kind != Value::Tab) { slot.kind = Value::Tab; slot.tab = std::make_shared<Table>(); }
                        else if (last) fail("table '" + path[k] + "' defined twice");
                        t = slot.tab.get();
                    }
                }
                cur = t;
            } else {
                std::string key = parse_key();
                skip_ws_inline();
                if (eof() || s[i] != '=') fail("expected '=' after key '" + key + "'");
                i++;
                if (cur->count(key)) fail("duplicate key `" + key + "`");
                (*cur)[key] = parse_value();
            }
            skip_ws_inline();
            skip_comment();
            if (!eof() && s[i] != '\n' && s[i] != '\r') fail("expected newline");
        }
        return root;
    }
};

// ---------------------------------------------------------------- serde-like field extraction
const Value &field(const Table &t, const char *name)
{
    auto it = t.find(name);
    if (it == t.end()) throw ParseError(std::string("missing field `") + name + "`");
    return it->second;
}
float as_f32(const Value &v, const char *name)
{
    if (v.kind != Value::Num) throw ParseError(std::string("invalid type for `") + name + "`, expected a number");
    return (float)v.num;
}
void as_vec3(const Value &v, const char *name, float out[3])
{
    if (v.kind != Value::Arr || v.arr.size() != 3) throw ParseError(std::string("invalid type for `") + name + "`, expected an array of 3 numbers");
    for (int k = 0; k < 3; k++) out[k] = as_f32(v.arr[k], name);
}
std::string as_str(const Value &v, const char *name)
{
    if (v.kind != Value::Str) throw ParseError(std::string("invalid type for `") + name + "`, expected a string");
    return v.str;
}

std::string dirname_of(const std::string &p)
{
    size_t k = p.find_last_of('/');
    if (k == std::string::npos) return ".";
    if (k == 0) return "/";
    return p.substr(0, k);
}
bool read_file(const std::string &path, std::string &out, std::string &err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = std::strerror(errno); return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

// ---------------------------------------------------------------- OBJ (Mesh::load)
struct Mesh {
    std::vector<rsrt_vec3> vertices, normals;
    std::vector<rsrt_triangle> triangles; // indices relative to this mesh
};

struct Corner { long v, vt, vn; bool has_vn; };

Mesh load_obj(const std::string &src, uint32_t material_id)
{
    Mesh mesh;
    // running totals over the whole file (OBJ indices are global and 1-based) and the totals at the
    // start of the current object (wavefront_obj rebases indices per object; mesh.rs adds the
    // per-object offsets back, :37-38, :54-64)
    size_t file_v = 0, file_vn = 0, obj_v0 = 0, obj_vn0 = 0;
    size_t mesh_v0 = 0, mesh_vn0 = 0; // offset of the current object inside mesh.vertices / normals
    std::istringstream in(src);
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "o") {
            obj_v0 = file_v; obj_vn0 = file_vn;
            mesh_v0 = mesh.vertices.size(); mesh_vn0 = mesh.normals.size();
        } else if (tag == "v" || tag == "vn") {
            double x, y, z;
            if (!(ls >> x >> y >> z)) throw ParseError("OBJ line " + std::to_string(lineno) + ": expected three numbers");
            rsrt_vec3 p = {{(float)x, (float)y, (float)z}, 0.0f};
            if (tag == "v") { mesh.vertices.push_back(p); file_v++; }
            else { mesh.normals.push_back(p); file_vn++; }
        } else if (tag == "f") {
            std::vector<Corner> cs;
            std::string tok;
            while (ls >> tok) {
                Corner c = {0, 0, 0, false};
                const char *p = tok.c_str();
                char *e;
                c.v = strtol(p, &e, 10);
                if (e == p) throw ParseError("OBJ line " + std::to_string(lineno) + ": bad face corner '" + tok + "'");
                if (*e == '/') {
                    p = e + 1;
                    if (*p != '/') { c.vt = strtol(p, &e, 10); } else e = (char *)p;
                    if (*e == '/') { p = e + 1; c.vn = strtol(p, &e, 10); c.has_vn = (e != p); }
                }
                cs.push_back(c);
            }
            if (cs.size() < 3) continue; // points / lines are dropped (mesh.rs:70)
            auto rel = [&](const Corner &c, uint32_t &vi, uint32_t &ni) {
                if (!c.has_vn) throw ParseError("Object must include baked normals"); // mesh.rs:60
                long v = c.v > 0 ? c.v - 1 - (long)obj_v0 : (long)(file_v - obj_v0) + c.v;
                long n = c.vn > 0 ? c.vn - 1 - (long)obj_vn0 : (long)(file_vn - obj_vn0) + c.vn;
                if (v < 0 || n < 0 || (size_t)v >= file_v - obj_v0 || (size_t)n >= file_vn - obj_vn0)
                    throw ParseError("OBJ line " + std::to_string(lineno) + ": index out of range");
                vi = (uint32_t)(mesh_v0 + (size_t)v);
                ni = (uint32_t)(mesh_vn0 + (size_t)n);
            };
            for (size_t k = 1; k + 1 < cs.size(); k++) { // fan (v0, vk, vk+1)
                rsrt_triangle t;
                rel(cs[0], t.vertex_0, t.normal_0);
                rel(cs[k], t.vertex_1, t.normal_1);
                rel(cs[k + 1], t.vertex_2, t.normal_2);
                t.material_id = material_id; // OBJ usemtl ignored (mesh.rs:66-67)
                mesh.triangles.push_back(t);
            }
        }
        // mtllib, usemtl, vt, s, g, l, p: ignored
    }
    return mesh;
}

// cgmath Deg<f32> -> Rad<f32>: deg * (pi/180 as f32) (scene.rs:305-314)
float deg_to_rad(float d) { return d * (float)(3.14159265358979323846 / 180.0); }

std::unique_ptr<rsrt_scene> build_scene(const Table &root, const std::string &path)
{
    auto sc = std::make_unique<rsrt_scene>();
    const Value &mats = field(root, "material");
    const Value &objs = field(root, "object");
    const Value &cam = field(root, "camera");
    if (mats.kind != Value::TabArr) throw ParseError("invalid type for `material`, expected an array of tables");
    if (objs.kind != Value::TabArr) throw ParseError("invalid type for `object`, expected an array of tables");
    if (cam.kind != Value::Tab) throw ParseError("invalid type for `camera`, expected a table");

    std::vector<std::string> names;
    for (auto &m : mats.tabs) {
        rsrt_material mm;
        std::memset(&mm, 0, sizeof mm);
        names.push_back(as_str(field(*m, "name"), "name"));
        as_vec3(field(*m, "color"), "color", mm.color);
        mm.roughness = as_f32(field(*m, "roughness"), "roughness");
        mm.metallic = as_f32(field(*m, "metallic"), "metallic");
        as_vec3(field(*m, "emission"), "emission", mm.emission);
        sc->materials.push_back(mm);
    }
    auto material_index = [&](const std::string &n) -> int { // first match wins (scene.rs:326-332)
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
        return -1;
    };
    auto object_error = [&](size_t index, const char *type, const std::string &msg) { // scene.rs:334-342
        return ParseError("Error in object " + std::to_string(index) + " (" + type + "): " + msg + "\n  --> " + path);
    };
    std::vector<Mesh> meshes;
    size_t index = 0;
    for (auto &o : objs.tabs) {
        if (o->size() != 1) throw ParseError("object " + std::to_string(index) + ": expected exactly one of `Sphere`, `Plane`, `Mesh`");
        const std::string &kind = o->begin()->first;
        const Value &body = o->begin()->second;
        if (kind != "Sphere" && kind != "Plane" && kind != "Mesh")
            throw ParseError("unknown variant `" + kind + "`, expected one of `Sphere`, `Plane`, `Mesh`");
        if (body.kind != Value::Tab) throw ParseError("invalid type for `" + kind + "`, expected a table");
        const Table &t = *body.tab;
        std::string mat = as_str(field(t, "material"), "material");
        int mid = material_index(mat);
        if (mid < 0) throw object_error(index, kind.c_str(), "Material '" + mat + "' does not exist."); // scene.rs:344-351
        if (kind == "Sphere") {
            rsrt_sphere s;
            std::memset(&s, 0, sizeof s);
            as_vec3(field(t, "pos"), "pos", s.pos);
            s.radius = as_f32(field(t, "radius"), "radius");
            s.material_id = (uint32_t)mid;
            sc->spheres.push_back(s);
        } else if (kind == "Plane") {
            rsrt_plane_desc p;
            as_vec3(field(t, "pos"), "pos", p.pos);
            as_vec3(field(t, "forward"), "forward", p.forward);
            as_vec3(field(t, "right"), "right", p.right);
            p.material_id = (uint32_t)mid;
            sc->plane_descs.push_back(p);
        } else {
            std::string rel = as_str(field(t, "path"), "path");
            std::string full = dirname_of(path) + "/" + rel, content, err;
            if (!read_file(full, content, err)) throw object_error(index, "Mesh", "Cannot open '" + rel + "': " + err);
            try {
                meshes.push_back(load_obj(content, (uint32_t)mid));
            } catch (const ParseError &e) {
                throw object_error(index, "Mesh", e.what());
            }
        }
        index++;
    }
    // PackedMeshes::pack_meshes (mesh.rs:91-113)
    for (const Mesh &m : meshes) {
        uint32_t vo = (uint32_t)sc->vertices.size(), no = (uint32_t)sc->normals.size();
        for (rsrt_triangle t : m.triangles) {
            t.vertex_0 += vo; t.vertex_1 += vo; t.vertex_2 += vo;
            t.normal_0 += no; t.normal_1 += no; t.normal_2 += no;
            sc->triangles.push_back(t);
        }
        sc->vertices.insert(sc->vertices.end(), m.vertices.begin(), m.vertices.end());
        sc->normals.insert(sc->normals.end(), m.normals.begin(), m.normals.end());
    }
    const Table &ct = *cam.tab;
    as_vec3(field(ct, "pos"), "pos", sc->camera.pos);
    sc->camera.yaw = deg_to_rad(as_f32(field(ct, "yaw"), "yaw"));
    sc->camera.pitch = deg_to_rad(as_f32(field(ct, "pitch"), "pitch"));
    sc->camera.fov_y = deg_to_rad(as_f32(field(ct, "fov_y"), "fov_y"));

    sc->planes.resize(sc->plane_descs.size());
    for (size_t i = 0; i < sc->plane_descs.size(); i++) rsrt_plane_to_uniform(&sc->plane_descs[i], &sc->planes[i]);
    int rc = rsrt_host::build_bvh_from_arrays(sc->spheres.data(), (uint32_t)sc->spheres.size(), sc->plane_descs.data(),
                                              (uint32_t)sc->plane_descs.size(), sc->vertices.data(), (uint32_t)sc->vertices.size(),
                                              sc->triangles.data(), (uint32_t)sc->triangles.size(), sc->primitives, sc->nodes,
                                              sc->depth);
    if (rc) throw ParseError("scene has no primitives");
    return sc;
}

void set_err(char *err, size_t n, const std::string &msg)
{
    if (err && n) { std::snprintf(err, n, "%s", msg.c_str()); }
}

} // namespace

extern "C" {

int rsrt_scene_load_toml(const char *path, rsrt_scene **out, char *err, size_t err_len)
{
    if (!path || !out) { set_err(err, err_len, "null argument"); return 1; }
    *out = nullptr;
    std::string content, ioerr;
    if (!read_file(path, content, ioerr)) { // scene.rs:236-242
        set_err(err, err_len, std::string("Couldn't open scene ") + path + ":\n  " + ioerr);
        return 2;
    }
    try {
        Table root;
        try {
            TomlParser p(content);
            root = p.parse();
            // field presence / type errors are parse errors too (serde), scene.rs:243-249
            *out = build_scene(root, path).release();
        } catch (const ParseError &e) {
            std::string m = e.what();
            if (m.rfind("Error in object", 0) == 0) set_err(err, err_len, m);
            else set_err(err, err_len, std::string("Couldn't parse scene ") + path + ":\n  " + m);
            return 3;
        }
    } catch (const std::exception &e) {
        set_err(err, err_len, e.what());
        return 4;
    }
    return 0;
}

void rsrt_scene_free(rsrt_scene *s) { delete s; }

void rsrt_scene_get_counts(const rsrt_scene *s, rsrt_scene_counts *o)
{
    o->n_materials = (uint32_t)s->materials.size();
    o->n_spheres = (uint32_t)s->spheres.size();
    o->n_planes = (uint32_t)s->planes.size();
    o->n_vertices = (uint32_t)s->vertices.size();
    o->n_normals = (uint32_t)s->normals.size();
    o->n_triangles = (uint32_t)s->triangles.size();
    o->n_primitives = (uint32_t)s->primitives.size();
    o->n_bvh_nodes = (uint32_t)s->nodes.size();
    o->bvh_depth = s->depth;
}
const rsrt_material *rsrt_scene_materials(const rsrt_scene *s) { return s->materials.data(); }
const rsrt_sphere *rsrt_scene_spheres(const rsrt_scene *s) { return s->spheres.data(); }
const rsrt_plane_desc *rsrt_scene_plane_descs(const rsrt_scene *s) { return s->plane_descs.data(); }
const rsrt_plane *rsrt_scene_planes(const rsrt_scene *s) { return s->planes.data(); }
const rsrt_vec3 *rsrt_scene_vertices(const rsrt_scene *s) { return s->vertices.data(); }
const rsrt_vec3 *rsrt_scene_normals(const rsrt_scene *s) { return s->normals.data(); }
const rsrt_triangle *rsrt_scene_triangles(const rsrt_scene *s) { return s->triangles.data(); }
const rsrt_primitive_info *rsrt_scene_primitives(const rsrt_scene *s) { return s->primitives.data(); }
const rsrt_bvh_node *rsrt_scene_bvh_nodes(const rsrt_scene *s) { return s->nodes.data(); }
void rsrt_scene_get_camera(const rsrt_scene *s, rsrt_camera_desc *o) { *o = s->camera; }

} // extern "C"
