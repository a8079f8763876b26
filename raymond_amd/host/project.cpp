// Project file and tile-message formats of the reference on the C++ side.
//   core/src/project.rs:13-57   serde-JSON `{"objects":[{"geometry":G,"material":M},...]}` with externally tagged enums
//   server/src/protocol.rs:9-14 adjacently tagged `{"type":"TileProgressed"|"TileFinished","data":Tile}`
// The JSON reader below covers the grammar serde_json accepts for these types (objects, arrays, numbers, strings with
// the standard escapes, true/false/null); cgmath's Vector3 deserialises from {"x":..,"y":..,"z":..} or [x, y, z].
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

#include "raymond.hpp"

namespace raymond {
namespace {

struct Json {
	enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
	bool b = false;
	double num = 0.0;
	std::string str;
	std::vector<Json> items;                             // Array
	std::vector<std::pair<std::string, Json>> members; // Object, in document order
	const Json *find(const std::string &key) const {
		for (const auto &m : members)
			if (m.first == key) return &m.second;
		return nullptr;
	}
};

struct Parser {
	const std::string &s;
	size_t i = 0;
	[[noreturn]] void fail(const std::string &what) const { throw Error(RMD_ERR_INVALID_ARGUMENT, "project JSON: " + what + " at offset " + std::to_string(i)); }
	void ws() {
		while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
	}
	bool eat(char c) {
		ws();
		if (i < s.size() && s[i] == c) {
			i++;
			return true;
		}
		return false;
	}
	void expect(char c) {
		if (!eat(c)) fail(std::string("expected '") + c + "'");
	}
	std::string string() {
		expect('"');
		std::string out;
		while (i < s.size() && s[i] != '"') {
			char c = s[i++];
			if (c != '\\') {
				out += c;
				continue;
			}
			if (i >= s.size()) fail("unterminated escape");
			char e = s[i++];
			switch (e) {
			case '"': out += '"'; break;
			case '\\': out += '\\'; break;
			case '/': out += '/'; break;
			case 'b': out += '\b'; break;
			case 'f': out += '\f'; break;
			case 'n': out += '\n'; break;
			case 'r': out += '\r'; break;
			case 't': out += '\t'; break;
			case 'u': {
				if (i + 4 > s.size()) fail("short \\u escape");
				unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
				i += 4;
				if (cp < 0x80) out += (char)cp; // paths in project files are ASCII in practice; encode the BMP as UTF-8
				else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
				else out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
			} break;
			default: fail("unknown escape");
			}
		}
		if (i >= s.size()) fail("unterminated string");
		i++;
		return out;
	}
	Json value() {
		ws();
		if (i >= s.size()) fail("unexpected end");
		Json v;
		char c = s[i];
		if (c == '{') {
			i++;
			v.kind = Json::Object;
			if (eat('}')) return v;
			do {
				ws();
				std::string k = string();
				expect(':');
				v.members.emplace_back(std::move(k), value());
			} while (eat(','));
			expect('}');
		} else if (c == '[') {
			i++;
			v.kind = Json::Array;
			if (eat(']')) return v;
			do v.items.push_back(value());
			while (eat(','));
			expect(']');
		} else if (c == '"') {
			v.kind = Json::String;
			v.str = string();
		} else if (s.compare(i, 4, "true") == 0) {
			i += 4, v.kind = Json::Bool, v.b = true;
		} else if (s.compare(i, 5, "false") == 0) {
			i += 5, v.kind = Json::Bool;
		} else if (s.compare(i, 4, "null") == 0) {
			i += 4;
		} else {
			const char *begin = s.c_str() + i;
			char *end = nullptr;
			v.num = std::strtod(begin, &end);
			if (end == begin) fail("unexpected character");
			v.kind = Json::Number;
			i += (size_t)(end - begin);
		}
		return v;
	}
};

double number(const Json &j, const char *what) {
	if (j.kind != Json::Number) throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: ") + what + " must be a number");
	return j.num;
}
Vector3 vec3(const Json &j, const char *what) {
	if (j.kind == Json::Object) {
		const Json *x = j.find("x"), *y = j.find("y"), *z = j.find("z");
		if (!x || !y || !z) throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: ") + what + " needs x, y, z");
		return {number(*x, what), number(*y, what), number(*z, what)};
	}
	if (j.kind == Json::Array && j.items.size() == 3) return {number(j.items[0], what), number(j.items[1], what), number(j.items[2], what)};
	throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: ") + what + " must be a Vector3");
}
const std::pair<std::string, Json> &variant(const Json &j, const char *what) {
	if (j.kind != Json::Object || j.members.size() != 1) throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: ") + what + " must be an object with exactly one variant key");
	return j.members[0];
}
const Json &field(const Json &j, const char *key) {
	const Json *f = j.kind == Json::Object ? j.find(key) : nullptr;
	if (!f) throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: missing field `") + key + "`"); // serde's wording
	return *f;
}
const Json &element(const Json &j, size_t k, size_t n, const char *what) {
	if (j.kind != Json::Array || j.items.size() != n) throw Error(RMD_ERR_INVALID_ARGUMENT, std::string("project JSON: ") + what + " must be an array of " + std::to_string(n));
	return j.items[k];
}

std::string num(double v) {
	char buf[40];
	std::snprintf(buf, sizeof buf, "%.17g", v);
	std::string s = buf;
	if (s.find_first_of(".eEn") == std::string::npos) s += ".0"; // serde_json writes floats with a fraction
	return s;
}
std::string vec_json(const double *v) { return "{\"x\":" + num(v[0]) + ",\"y\":" + num(v[1]) + ",\"z\":" + num(v[2]) + "}"; }
std::string quoted(const std::string &s) {
	std::string out = "\"";
	for (char c : s) {
		if (c == '"' || c == '\\') out += '\\', out += c;
		else if (c == '\n') out += "\\n";
		else out += c;
	}
	return out + "\"";
}

} // namespace

Project Project::loads(const std::string &text, const std::string &base_dir) {
	Parser p{text};
	Json doc = p.value();
	p.ws();
	if (p.i != text.size()) p.fail("trailing characters");
	const Json &objs = field(doc, "objects");
	if (objs.kind != Json::Array) throw Error(RMD_ERR_INVALID_ARGUMENT, "project JSON: `objects` must be an array");
	Project out;
	out.base_dir = base_dir;
	for (const Json &o : objs.items) {
		ProjectObject po;
		const auto &g = variant(field(o, "geometry"), "geometry");
		if (g.first == "Plane") {
			po.kind = ProjectObject::PlaneGeometry;
			po.plane = Plane{vec3(field(g.second, "origin"), "Plane.origin"), vec3(field(g.second, "normal"), "Plane.normal")};
		} else if (g.first == "Sphere") {
			po.kind = ProjectObject::SphereGeometry;
			po.sphere = Sphere{vec3(field(g.second, "origin"), "Sphere.origin"), number(field(g.second, "radius"), "Sphere.radius")};
		} else if (g.first == "Mesh") {
			if (g.second.kind != Json::String) throw Error(RMD_ERR_INVALID_ARGUMENT, "project JSON: Mesh must be a path string");
			po.kind = ProjectObject::MeshGeometry;
			po.mesh_path = g.second.str;
		} else {
			throw Error(RMD_ERR_INVALID_ARGUMENT, "project JSON: unknown variant `" + g.first + "`, expected one of `Plane`, `Sphere`, `Mesh`");
		}
		const auto &m = variant(field(o, "material"), "material");
		if (m.first == "Diffuse") po.material = Material::Diffuse(vec3(element(m.second, 0, 2, "Diffuse"), "Diffuse.0"), number(element(m.second, 1, 2, "Diffuse"), "Diffuse.1"));
		else if (m.first == "Metal") po.material = Material::Metal(vec3(element(m.second, 0, 2, "Metal"), "Metal.0"), number(element(m.second, 1, 2, "Metal"), "Metal.1"));
		else if (m.first == "Emission")
			po.material = Material::Emission(vec3(element(m.second, 0, 4, "Emission"), "Emission.0"), vec3(element(m.second, 1, 4, "Emission"), "Emission.1"),
			                                 number(element(m.second, 2, 4, "Emission"), "Emission.2"), number(element(m.second, 3, 4, "Emission"), "Emission.3"));
		else throw Error(RMD_ERR_INVALID_ARGUMENT, "project JSON: unknown variant `" + m.first + "`, expected one of `Diffuse`, `Metal`, `Emission`");
		out.objects.push_back(std::move(po));
	}
	return out;
}

Project Project::load(const std::string &path) { // project.rs:33-36
	std::ifstream in(path);
	if (!in) throw Error(RMD_ERR_INVALID_ARGUMENT, "Project::load: cannot open " + path);
	std::stringstream ss;
	ss << in.rdbuf();
	const size_t slash = path.find_last_of('/');
	return loads(ss.str(), slash == std::string::npos ? std::string() : path.substr(0, slash));
}

std::string Project::dumps() const {
	std::string out = "{\"objects\":[";
	for (size_t i = 0; i < objects.size(); i++) {
		const ProjectObject &o = objects[i];
		if (i) out += ",";
		out += "{\"geometry\":";
		if (o.kind == ProjectObject::PlaneGeometry) out += "{\"Plane\":{\"origin\":" + vec_json(o.plane.origin.data()) + ",\"normal\":" + vec_json(o.plane.normal.data()) + "}}";
		else if (o.kind == ProjectObject::SphereGeometry) out += "{\"Sphere\":{\"origin\":" + vec_json(o.sphere.origin.data()) + ",\"radius\":" + num(o.sphere.radius) + "}}";
		else out += "{\"Mesh\":" + quoted(o.mesh_path) + "}";
		out += ",\"material\":";
		const Material &m = o.material;
		if (m.kind == RMD_MAT_DIFFUSE) out += "{\"Diffuse\":[" + vec_json(m.color.data()) + "," + num(m.roughness) + "]}";
		else if (m.kind == RMD_MAT_METAL) out += "{\"Metal\":[" + vec_json(m.color.data()) + "," + num(m.roughness) + "]}";
		else out += "{\"Emission\":[" + vec_json(m.color.data()) + "," + vec_json(m.aux.data()) + "," + num(m.aux[3]) + "," + num(m.aux[4]) + "]}";
		out += "}";
	}
	return out + "]}";
}

Scene Project::build_scene() const { // project.rs:38-57
	Scene scene;
	std::map<std::string, std::shared_ptr<AccGrid>> built; // one grid per file, shared like an Arc
	for (const ProjectObject &o : objects) {
		Geometry g;
		if (o.kind == ProjectObject::PlaneGeometry) g = Geometry::Plane_(o.plane);
		else if (o.kind == ProjectObject::SphereGeometry) g = Geometry::Sphere_(o.sphere);
		else {
			const std::string path = (!o.mesh_path.empty() && o.mesh_path[0] == '/') || base_dir.empty() ? o.mesh_path : base_dir + "/" + o.mesh_path;
			auto it = built.find(path);
			if (it == built.end()) it = built.emplace(path, AccGrid::build_from_mesh(Mesh::load_ply(path))).first;
			g = Geometry::Grid(it->second);
		}
		scene.objects.push_back(Object{g, o.material});
	}
	return scene;
}

std::string message_to_json(const Message &message) {
	const Tile &t = message.tile;
	std::string out = std::string("{\"type\":\"") + (message.kind == Message::TileFinished ? "TileFinished" : "TileProgressed") + "\",\"data\":{";
	out += "\"sample_count\":" + std::to_string(t.sample_count) + ",\"width\":" + std::to_string(t.width) + ",\"height\":" + std::to_string(t.height) +
	       ",\"left\":" + std::to_string(t.left) + ",\"top\":" + std::to_string(t.top) + ",\"data\":[";
	for (size_t i = 0; i < t.data.size(); i++) out += (i ? "," : "") + vec_json(t.data[i].data());
	return out + "]}}";
}

} // namespace raymond
