// Minimal JSON DOM with the subset of the jsoncpp 1.8.0 interface the
// reference's scene loader uses (reference: src/world_load.cpp:41-115,576-589,
// src/jsonResolve.cpp:5-48, src/camera.cpp:15-48 call Json::Value::isArray /
// isObject / size / operator[] / asFloat / asInt / asBool / asString /
// asCString / getMemberNames and Json::Reader::parse /
// getFormattedErrorMessages). jsoncpp itself is not vendored; this is a fresh
// implementation with the same observable behaviour on those calls:
//   * operator[] on a missing key / out-of-range index yields a null Value;
//   * null converts to 0 / 0.0f / false / "";
//   * integers without '.', 'e' or 'E' are stored as integers, others as double,
//     asFloat() is static_cast<float>(stored value);
//   * object members iterate in sorted key order (jsoncpp keeps a std::map);
//   * C and C++ comments are accepted by the reader.
#pragma once
#include <istream>
#include <map>
#include <string>
#include <vector>

namespace Json {

enum ValueType { nullValue = 0, intValue, realValue, stringValue, booleanValue, arrayValue, objectValue };

class Value {
public:
	typedef std::vector<std::string> Members;

	Value() : type_(nullValue), int_(0), real_(0.0), bool_(false) {}
	explicit Value(ValueType t) : type_(t), int_(0), real_(0.0), bool_(false) {}

	ValueType type() const { return type_; }
	bool isNull() const { return type_ == nullValue; }
	bool isArray() const { return type_ == arrayValue; }
	bool isObject() const { return type_ == objectValue; }
	bool isString() const { return type_ == stringValue; }
	bool isNumeric() const { return type_ == intValue || type_ == realValue; }
	bool isBool() const { return type_ == booleanValue; }
	bool isMember(const std::string& key) const { return type_ == objectValue && object_.count(key) != 0; }

	unsigned size() const;
	const Value& operator[](int index) const;
	const Value& operator[](unsigned index) const { return (*this)[(int)index]; }
	const Value& operator[](const char* key) const;
	const Value& operator[](const std::string& key) const { return (*this)[key.c_str()]; }

	float asFloat() const;
	double asDouble() const;
	int asInt() const;
	bool asBool() const;
	std::string asString() const;
	const char* asCString() const { return string_.c_str(); }
	Members getMemberNames() const;

	// builders (used by the reader and by tests)
	static Value makeInt(long long v) { Value r(intValue); r.int_ = v; return r; }
	static Value makeReal(double v) { Value r(realValue); r.real_ = v; return r; }
	static Value makeBool(bool v) { Value r(booleanValue); r.bool_ = v; return r; }
	static Value makeString(const std::string& s) { Value r(stringValue); r.string_ = s; return r; }
	void append(const Value& v) { type_ = arrayValue; array_.push_back(v); }
	void set(const std::string& key, const Value& v) { type_ = objectValue; object_[key] = v; }

private:
	ValueType type_;
	long long int_;
	double real_;
	bool bool_;
	std::string string_;
	std::vector<Value> array_;
	std::map<std::string, Value> object_;
};

class Reader {
public:
	// collectComments is accepted for signature compatibility; comments are always skipped.
	bool parse(std::istream& is, Value& root, bool collectComments = true);
	bool parse(const std::string& document, Value& root, bool collectComments = true);
	std::string getFormattedErrorMessages() const { return error_; }

private:
	std::string error_;
};

} // namespace Json
