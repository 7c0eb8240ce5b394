// Recursive-descent JSON reader behind the jsoncpp-shaped interface in json.h.
#include "json/json.h"

#include <cstdlib>
#include <cstring>
#include <iterator>
#include <sstream>

namespace Json {

static const Value kNull;

unsigned Value::size() const {
	if (type_ == arrayValue) return (unsigned)array_.size();
	if (type_ == objectValue) return (unsigned)object_.size();
	return 0;
}

const Value& Value::operator[](int index) const {
	if (type_ != arrayValue || index < 0 || index >= (int)array_.size()) return kNull;
	return array_[index];
}

const Value& Value::operator[](const char* key) const {
	if (type_ != objectValue) return kNull;
	std::map<std::string, Value>::const_iterator it = object_.find(key);
	return it == object_.end() ? kNull : it->second;
}

double Value::asDouble() const {
	switch (type_) {
		case intValue: return (double)int_;
		case realValue: return real_;
		case booleanValue: return bool_ ? 1.0 : 0.0;
		default: return 0.0;
	}
}

float Value::asFloat() const {
	switch (type_) {
		case intValue: return static_cast<float>(int_);
		case realValue: return static_cast<float>(real_);
		case booleanValue: return bool_ ? 1.0f : 0.0f;
		default: return 0.0f;
	}
}

int Value::asInt() const {
	switch (type_) {
		case intValue: return (int)int_;
		case realValue: return (int)real_;
		case booleanValue: return bool_ ? 1 : 0;
		default: return 0;
	}
}

bool Value::asBool() const {
	switch (type_) {
		case booleanValue: return bool_;
		case intValue: return int_ != 0;
		case realValue: return real_ != 0.0;
		default: return false;
	}
}

std::string Value::asString() const {
	switch (type_) {
		case stringValue: return string_;
		case booleanValue: return bool_ ? "true" : "false";
		case intValue: { std::ostringstream o; o << int_; return o.str(); }
		case realValue: { std::ostringstream o; o << real_; return o.str(); }
		default: return "";
	}
}

Value::Members Value::getMemberNames() const {
	Members m;
	if (type_ == objectValue)
		for (std::map<std::string, Value>::const_iterator it = object_.begin(); it != object_.end(); ++it)
			m.push_back(it->first);
	return m;
}

namespace {

struct Parser {
	const char* p;
	const char* begin;
	const char* end;
	std::string err;

	bool fail(const std::string& msg) {
		if (err.empty()) {
			int line = 1, col = 1;
			for (const char* q = begin; q < p && q < end; ++q) {
				if (*q == '\n') { line++; col = 1; } else col++;
			}
			std::ostringstream o;
			o << "* Line " << line << ", Column " << col << "\n  " << msg << "\n";
			err = o.str();
		}
		return false;
	}

	void skipWs() {
		for (;;) {
			while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
			if (p + 1 < end && p[0] == '/' && p[1] == '/') {
				while (p < end && *p != '\n') ++p;
			} else if (p + 1 < end && p[0] == '/' && p[1] == '*') {
				p += 2;
				while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p;
				p = (p + 1 < end) ? p + 2 : end;
			} else {
				return;
			}
		}
	}

	static void appendUtf8(std::string& out, unsigned cp) {
		if (cp < 0x80) out += (char)cp;
		else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
		else if (cp < 0x10000) {
			out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
		} else {
			out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
			out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
		}
	}

	bool parseHex4(unsigned& v) {
		if (end - p < 4) return fail("Bad unicode escape sequence in string: four digits expected.");
		v = 0;
		for (int i = 0; i < 4; ++i) {
			char c = *p++;
			v <<= 4;
			if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
			else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
			else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
			else return fail("Bad unicode escape sequence in string: hexadecimal digit expected.");
		}
		return true;
	}

	bool parseString(std::string& out) {
		++p; // opening quote
		while (p < end && *p != '"') {
			char c = *p++;
			if (c != '\\') { out += c; continue; }
			if (p >= end) return fail("Empty escape sequence in string");
			char e = *p++;
			switch (e) {
				case '"': out += '"'; break;
				case '/': out += '/'; break;
				case '\\': out += '\\'; break;
				case 'b': out += '\b'; break;
				case 'f': out += '\f'; break;
				case 'n': out += '\n'; break;
				case 'r': out += '\r'; break;
				case 't': out += '\t'; break;
				case 'u': {
					unsigned cp = 0;
					if (!parseHex4(cp)) return false;
					if (cp >= 0xD800 && cp <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
						p += 2;
						unsigned lo = 0;
						if (!parseHex4(lo)) return false;
						cp = 0x10000 + ((cp & 0x3FF) << 10) + (lo & 0x3FF);
					}
					appendUtf8(out, cp);
					break;
				}
				default: return fail("Bad escape sequence in string");
			}
		}
		if (p >= end) return fail("Missing '\"' at end of string");
		++p; // closing quote
		return true;
	}

	bool parseNumber(Value& out) {
		const char* s = p;
		bool isReal = false;
		if (p < end && (*p == '-' || *p == '+')) ++p;
		while (p < end) {
			char c = *p;
			if (c >= '0' && c <= '9') { ++p; continue; }
			if (c == '.' || c == 'e' || c == 'E') { isReal = true; ++p; continue; }
			if ((c == '+' || c == '-') && (p[-1] == 'e' || p[-1] == 'E')) { ++p; continue; }
			break;
		}
		std::string tok(s, p);
		if (tok.empty() || tok == "-" || tok == "+") { p = s; return fail("Syntax error: value, object or array expected."); }
		char* ep = 0;
		if (!isReal) {
			long long v = strtoll(tok.c_str(), &ep, 10);
			if (*ep == 0) { out = Value::makeInt(v); return true; }
		}
		double d = strtod(tok.c_str(), &ep);
		if (*ep != 0) { p = s; return fail("'" + tok + "' is not a number."); }
		out = Value::makeReal(d);
		return true;
	}

	bool parseValue(Value& out, int depth) {
		if (depth > 1000) return fail("Exceeded stackLimit in readValue().");
		skipWs();
		if (p >= end) return fail("Syntax error: value, object or array expected.");
		char c = *p;
		if (c == '{') {
			++p;
			out = Value(objectValue);
			skipWs();
			if (p < end && *p == '}') { ++p; return true; }
			for (;;) {
				skipWs();
				if (p >= end || *p != '"') return fail("Missing '}' or object member name");
				std::string key;
				if (!parseString(key)) return false;
				skipWs();
				if (p >= end || *p != ':') return fail("Missing ':' after object member name");
				++p;
				Value v;
				if (!parseValue(v, depth + 1)) return false;
				out.set(key, v);
				skipWs();
				if (p < end && *p == ',') { ++p; continue; }
				if (p < end && *p == '}') { ++p; return true; }
				return fail("Missing ',' or '}' in object declaration");
			}
		}
		if (c == '[') {
			++p;
			out = Value(arrayValue);
			skipWs();
			if (p < end && *p == ']') { ++p; return true; }
			for (;;) {
				Value v;
				if (!parseValue(v, depth + 1)) return false;
				out.append(v);
				skipWs();
				if (p < end && *p == ',') { ++p; continue; }
				if (p < end && *p == ']') { ++p; return true; }
				return fail("Missing ',' or ']' in array declaration");
			}
		}
		if (c == '"') {
			std::string s;
			if (!parseString(s)) return false;
			out = Value::makeString(s);
			return true;
		}
		if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; out = Value::makeBool(true); return true; }
		if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; out = Value::makeBool(false); return true; }
		if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; out = Value(); return true; }
		if (c == '-' || c == '+' || (c >= '0' && c <= '9')) return parseNumber(out);
		return fail("Syntax error: value, object or array expected.");
	}
};

} // namespace

bool Reader::parse(const std::string& document, Value& root, bool) {
	Parser ps;
	ps.begin = ps.p = document.data();
	ps.end = document.data() + document.size();
	error_.clear();
	Value v;
	bool ok = ps.parseValue(v, 0);
	if (ok) {
		ps.skipWs();
		if (ps.p != ps.end) ok = ps.fail("Extra non-whitespace after JSON value.");
	}
	if (ok && !v.isArray() && !v.isObject())
		ok = ps.fail("A valid JSON document must be either an array or an object value.");
	if (!ok) { error_ = ps.err; return false; }
	root = v;
	return true;
}

bool Reader::parse(std::istream& is, Value& root, bool collectComments) {
	std::string doc((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
	return parse(doc, root, collectComments);
}

} // namespace Json
