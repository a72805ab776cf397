// XML scene loader over the C ABI: the role of src/loader_xml/loader_xml.cc (main, :75-316) and the
// SAX state machine of src/common/import_xml.cc (:330-780), without libxml2 — YafaRay scene files
// use only elements and attributes (no text nodes, entities only in attribute values), so a small
// hand-written tokenizer is enough.  Element handling follows startElScene__ (:378-545),
// startElMesh__ (:550-638), startElParammap__/endElParammap__ (:676-737) and parseParam__ (:273-318).
#include "../../include/yafaray_c_api.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

extern "C" void yafaray_internal_set_error(yafaray_interface_t *yi, const char *msg); // yafaray_c_api.cpp
extern "C" void yafaray_internal_set_base_dir(yafaray_interface_t *yi, const char *dir);

namespace {

typedef std::vector<std::pair<std::string, std::string>> Attrs;

struct Tok { enum Kind { Open, Close, SelfClose, End } kind = End; std::string name; Attrs attrs; };

struct Lexer
{
	const std::string &s; size_t i = 0; std::string error;
	explicit Lexer(const std::string &src) : s(src) {}
	static std::string unescape(const std::string &v)
	{
		std::string o; o.reserve(v.size());
		for(size_t k = 0; k < v.size(); ++k)
		{
			if(v[k] == '&')
			{
				if(!v.compare(k, 5, "&amp;")) { o += '&'; k += 4; continue; }
				if(!v.compare(k, 4, "&lt;")) { o += '<'; k += 3; continue; }
				if(!v.compare(k, 4, "&gt;")) { o += '>'; k += 3; continue; }
				if(!v.compare(k, 6, "&quot;")) { o += '"'; k += 5; continue; }
				if(!v.compare(k, 6, "&apos;")) { o += '\''; k += 5; continue; }
			}
			o += v[k];
		}
		return o;
	}
	Tok next()
	{
		Tok t;
		for(;;)
		{
			const size_t lt = s.find('<', i);
			if(lt == std::string::npos) { t.kind = Tok::End; return t; }
			i = lt;
			if(!s.compare(i, 4, "<!--")) { const size_t e = s.find("-->", i + 4); if(e == std::string::npos) { error = "unterminated comment"; return t; } i = e + 3; continue; }
			if(!s.compare(i, 2, "<?")) { const size_t e = s.find("?>", i + 2); if(e == std::string::npos) { error = "unterminated declaration"; return t; } i = e + 2; continue; }
			if(!s.compare(i, 2, "<!")) { const size_t e = s.find('>', i + 2); if(e == std::string::npos) { error = "unterminated doctype"; return t; } i = e + 1; continue; }
			break;
		}
		++i;
		bool closing = false;
		if(i < s.size() && s[i] == '/') { closing = true; ++i; }
		size_t b = i;
		while(i < s.size() && !isspace((unsigned char)s[i]) && s[i] != '>' && s[i] != '/') ++i;
		t.name = s.substr(b, i - b);
		for(;;)
		{
			while(i < s.size() && isspace((unsigned char)s[i])) ++i;
			if(i >= s.size()) { error = "unexpected end of file in <" + t.name + ">"; t.kind = Tok::End; return t; }
			if(s[i] == '>') { ++i; t.kind = closing ? Tok::Close : Tok::Open; return t; }
			if(s[i] == '/' && i + 1 < s.size() && s[i + 1] == '>') { i += 2; t.kind = Tok::SelfClose; return t; }
			b = i;
			while(i < s.size() && s[i] != '=' && !isspace((unsigned char)s[i]) && s[i] != '>') ++i;
			std::string an = s.substr(b, i - b);
			while(i < s.size() && isspace((unsigned char)s[i])) ++i;
			if(i >= s.size() || s[i] != '=') { error = "attribute without value in <" + t.name + ">"; t.kind = Tok::End; return t; }
			++i;
			while(i < s.size() && isspace((unsigned char)s[i])) ++i;
			if(i >= s.size() || (s[i] != '"' && s[i] != '\'')) { error = "unquoted attribute in <" + t.name + ">"; t.kind = Tok::End; return t; }
			const char q = s[i++];
			b = i;
			while(i < s.size() && s[i] != q) ++i;
			if(i >= s.size()) { error = "unterminated attribute in <" + t.name + ">"; t.kind = Tok::End; return t; }
			t.attrs.emplace_back(an, unescape(s.substr(b, i - b)));
			++i;
		}
	}
};

// parseParam__, import_xml.cc:273-318
void set_param(yafaray_interface_t *yi, const std::string &name, const Attrs &a)
{
	if(a.empty()) return;
	if(a.size() == 1)
	{
		const std::string &k = a[0].first, &v = a[0].second;
		if(k == "ival") { yafaray_paramsSetInt(yi, name.c_str(), atoi(v.c_str())); return; }
		if(k == "fval") { yafaray_paramsSetFloat(yi, name.c_str(), atof(v.c_str())); return; }
		if(k == "bval") { yafaray_paramsSetBool(yi, name.c_str(), v == "true"); return; }
		if(k == "sval") { yafaray_paramsSetString(yi, name.c_str(), v.c_str()); return; }
	}
	double p[3] = {0, 0, 0}; float c[4] = {0, 0, 0, 0}; int type = 0;
	float m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // the reference leaves unset elements uninitialised; exporters write all 16
	for(const auto &kv : a)
	{
		const std::string &k = kv.first;
		if(k.size() == 3 && k[0] == 'm' && k[1] >= '0' && k[1] <= '3' && k[2] >= '0' && k[2] <= '3')
		{	// "mij", import_xml.cc:302-308
			m[4 * (k[1] - '0') + (k[2] - '0')] = (float)atof(kv.second.c_str()); type = 3;
			continue;
		}
		if(kv.first.size() != 1) continue;
		switch(kv.first[0])
		{
			case 'x': p[0] = atof(kv.second.c_str()); type = 1; break;
			case 'y': p[1] = atof(kv.second.c_str()); type = 1; break;
			case 'z': p[2] = atof(kv.second.c_str()); type = 1; break;
			case 'r': c[0] = (float)atof(kv.second.c_str()); type = 2; break;
			case 'g': c[1] = (float)atof(kv.second.c_str()); type = 2; break;
			case 'b': c[2] = (float)atof(kv.second.c_str()); type = 2; break;
			case 'a': c[3] = (float)atof(kv.second.c_str()); type = 2; break;
			default: break;
		}
	}
	if(type == 1) yafaray_paramsSetPoint(yi, name.c_str(), p[0], p[1], p[2]);
	else if(type == 3) yafaray_paramsSetMatrix(yi, name.c_str(), m, 0);
	else if(type == 2) yafaray_paramsSetColor(yi, name.c_str(), c[0], c[1], c[2], c[3]);
}

const std::string *attr(const Attrs &a, const char *k)
{
	for(const auto &kv : a) if(kv.first == k) return &kv.second;
	return nullptr;
}

} // namespace

static yafaray_bool_t load_xml_impl(yafaray_interface_t *yi, const char *path);
// no C++ exception (an allocation a damaged file provokes, a library error) may cross the C boundary
extern "C" yafaray_bool_t yafaray_loadXml(yafaray_interface_t *yi, const char *path)
{
	try { return load_xml_impl(yi, path); }
	catch(const std::exception &e) { yafaray_internal_set_error(yi, (std::string("loadXml: ") + e.what()).c_str()); }
	catch(...) { yafaray_internal_set_error(yi, "loadXml: unknown failure"); }
	return 0;
}
static yafaray_bool_t load_xml_impl(yafaray_interface_t *yi, const char *path)
{
	if(!yi || !path) return 0;
	std::string src;
	{
		FILE *f = std::fopen(path, "rb");
		if(!f) { yafaray_paramsClearAll(yi); return 0; }
		char buf[1 << 16]; size_t n;
		while((n = std::fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, n);
		std::fclose(f);
	}
	{	// texture file names in the scene are relative to where yafaray-xml is started, by convention the scene file's directory
		const std::string ps(path); const size_t slash = ps.rfind('/');
		yafaray_internal_set_base_dir(yi, slash == std::string::npos ? "." : ps.substr(0, slash).c_str());
	}
	Lexer lx(src);
	std::map<std::string, yafaray_material_t *> materials;
	std::vector<std::string> errors;
	bool in_scene = false;
	bool ok = true;
	for(;;)
	{
		Tok t = lx.next();
		if(t.kind == Tok::End) break;
		if(!in_scene)
		{
			if(t.kind == Tok::Open && t.name == "scene")
			{	// startElDocument__, :330-350
				const std::string *ty = attr(t.attrs, "type");
				if(!yafaray_startScene(yi, (ty && *ty == "universal") ? 1 : 0)) return 0;
				in_scene = true;
			}
			continue;
		}
		if(t.kind == Tok::Close) { if(t.name == "scene") in_scene = false; continue; }
		const std::string &el = t.name;
		if(el == "material" || el == "integrator" || el == "light" || el == "camera" || el == "background" ||
		   el == "texture" || el == "object" || el == "volumeregion" || el == "render_passes" || el == "logging_badge" || el == "render")
		{
			const std::string *name = attr(t.attrs, "name");
			yafaray_paramsClearAll(yi);
			if(t.kind == Tok::Open)
			{
				// parameter map until the matching end tag; <list_element> opens an extended ParamMap (:683-690)
				int depth = 0;
				for(;;)
				{
					Tok p = lx.next();
					if(p.kind == Tok::End) { ok = false; break; }
					if(p.kind == Tok::Close)
					{
						if(p.name == "list_element") { yafaray_paramsEndList(yi); --depth; continue; }
						if(p.name == el && depth == 0) break;
						continue;
					}
					if(p.name == "list_element") { yafaray_paramsPushList(yi); if(p.kind == Tok::Open) ++depth; else yafaray_paramsEndList(yi); continue; }
					set_param(yi, p.name, p.attrs);
				}
			}
			if(el == "render") { break; /* the render ParamMap stays current for yafaray_render (:526-530) */ }
			if(!name) { errors.push_back("<" + el + "> without a name"); ok = false; continue; }
			bool created = true;
			if(el == "material") { yafaray_material_t *m = yafaray_createMaterial(yi, name->c_str()); if(m) materials[*name] = m; else created = false; }
			else if(el == "integrator") created = yafaray_createIntegrator(yi, name->c_str()) != nullptr;
			else if(el == "light") created = yafaray_createLight(yi, name->c_str()) != nullptr;
			else if(el == "camera") created = yafaray_createCamera(yi, name->c_str()) != nullptr;
			else if(el == "background") created = yafaray_createBackground(yi, name->c_str()) != nullptr;
			else if(el == "texture") created = yafaray_createTexture(yi, name->c_str()) != nullptr;
			else if(el == "render_passes" || el == "logging_badge") created = true;   // only the combined pass exists here
			else { errors.push_back("<" + el + "> elements are outside the GPU path's scope"); created = false; }
			if(!created) { errors.push_back(std::string("<") + el + " name=\"" + *name + "\">: " + yafaray_getLastError(yi)); ok = false; }
		}
		else if(el == "mesh" && t.kind == Tok::Open)
		{	// :401-427, startElMesh__ :550-638
			int vertices = 0, triangles = 0, type = 0, id = -1, pass = 0; bool has_orco = false, has_uv = false;
			for(const auto &kv : t.attrs)
			{
				if(kv.first == "has_orco") has_orco = kv.second == "true";
				else if(kv.first == "has_uv") has_uv = kv.second == "true";
				else if(kv.first == "vertices") vertices = atoi(kv.second.c_str());
				else if(kv.first == "faces") triangles = atoi(kv.second.c_str());
				else if(kv.first == "type") type = atoi(kv.second.c_str());
				else if(kv.first == "id") id = atoi(kv.second.c_str());
				else if(kv.first == "obj_pass_index") pass = atoi(kv.second.c_str());
			}
			if(!yafaray_startGeometry(yi)) { errors.push_back(yafaray_getLastError(yi)); ok = false; }
			const unsigned int mid = id == -1 ? yafaray_getNextFreeId(yi) : (unsigned int)id;
			if(!yafaray_startTriMesh(yi, mid, vertices, triangles, has_orco, has_uv, type, pass)) { errors.push_back(yafaray_getLastError(yi)); ok = false; }
			const yafaray_material_t *mat = nullptr;
			for(;;)
			{
				Tok p = lx.next();
				if(p.kind == Tok::End) { ok = false; break; }
				if(p.kind == Tok::Close) { if(p.name == "mesh") break; continue; }
				if(p.name == "p")
				{	// parsePoint__, import_xml.cc:214-249: x y z and, for has_orco meshes, ox oy oz
					double x = 0, y = 0, z = 0, ox = 0, oy = 0, oz = 0;
					for(const auto &kv : p.attrs)
					{
						const std::string &k = kv.first;
						if(k.size() == 1) { if(k[0] == 'x') x = atof(kv.second.c_str()); else if(k[0] == 'y') y = atof(kv.second.c_str()); else if(k[0] == 'z') z = atof(kv.second.c_str()); }
						else if(k.size() == 2 && k[0] == 'o') { if(k[1] == 'x') ox = atof(kv.second.c_str()); else if(k[1] == 'y') oy = atof(kv.second.c_str()); else if(k[1] == 'z') oz = atof(kv.second.c_str()); }
					}
					if(has_orco) yafaray_addVertexWithOrco(yi, x, y, z, ox, oy, oz);
					else yafaray_addVertex(yi, x, y, z);
				}
				else if(p.name == "n")
				{
					double x = 0, y = 0, z = 0; int got = 0;
					for(const auto &kv : p.attrs) if(kv.first.size() == 1) { if(kv.first[0] == 'x') { x = atof(kv.second.c_str()); ++got; } else if(kv.first[0] == 'y') { y = atof(kv.second.c_str()); ++got; } else if(kv.first[0] == 'z') { z = atof(kv.second.c_str()); ++got; } }
					if(got == 3) yafaray_addNormal(yi, x, y, z);
				}
				else if(p.name == "f")
				{
					int a = 0, b = 0, c = 0, uv_a = 0, uv_b = 0, uv_c = 0;
					for(const auto &kv : p.attrs)
					{
						if(kv.first.size() == 1) { if(kv.first[0] == 'a') a = atoi(kv.second.c_str()); else if(kv.first[0] == 'b') b = atoi(kv.second.c_str()); else if(kv.first[0] == 'c') c = atoi(kv.second.c_str()); }
						else if(kv.first == "uv_a") uv_a = atoi(kv.second.c_str());
						else if(kv.first == "uv_b") uv_b = atoi(kv.second.c_str());
						else if(kv.first == "uv_c") uv_c = atoi(kv.second.c_str());
					}
					const bool added = mat && (has_uv ? yafaray_addTriangleWithUv(yi, a, b, c, uv_a, uv_b, uv_c, mat) : yafaray_addTriangle(yi, a, b, c, mat));
					if(!added) { if(ok) errors.push_back(mat ? yafaray_getLastError(yi) : "face before set_material / unknown material"); ok = false; }
				}
				else if(p.name == "uv")
				{	// :593-619
					float u = 0, v = 0;
					for(const auto &kv : p.attrs) if(!kv.first.empty()) { if(kv.first[0] == 'u') u = (float)atof(kv.second.c_str()); else if(kv.first[0] == 'v') v = (float)atof(kv.second.c_str()); }
					if(!std::isfinite(u)) u = 0.f;
					if(!std::isfinite(v)) v = 0.f;
					yafaray_addUv(yi, u, v);
				}
				else if(p.name == "set_material")
				{
					mat = nullptr;
					if(!p.attrs.empty()) { auto it = materials.find(p.attrs[0].second); if(it != materials.end()) mat = it->second; }
				}
			}
			if(!yafaray_endTriMesh(yi)) { errors.push_back(yafaray_getLastError(yi)); ok = false; }
			if(!yafaray_endGeometry(yi)) { errors.push_back(yafaray_getLastError(yi)); ok = false; }
		}
		else if(el == "smooth")
		{	// :429-444
			unsigned int id = 0; double angle = 181;
			for(const auto &kv : t.attrs) { if(kv.first == "ID") id = (unsigned int)atoi(kv.second.c_str()); else if(kv.first == "angle") angle = atof(kv.second.c_str()); }
			yafaray_startGeometry(yi);
			if(!yafaray_smoothMesh(yi, id, angle)) { errors.push_back(yafaray_getLastError(yi)); ok = false; }
			yafaray_endGeometry(yi);
		}
		else if(t.kind == Tok::Open)
		{
			errors.push_back("<" + el + "> elements are outside the GPU path's scope");
			ok = false;
		}
	}
	if(!lx.error.empty()) { errors.push_back("XML syntax: " + lx.error); ok = false; }
	if(!ok)
	{
		// leave the first diagnostic where yafaray_getLastError finds it: route it through a failing call
		std::string msg = "loadXml: ";
		for(size_t k = 0; k < errors.size() && k < 4; ++k) { if(k) msg += " | "; msg += errors[k]; }
		yafaray_internal_set_error(yi, msg.c_str());
		return 0;
	}
	return 1;
}
