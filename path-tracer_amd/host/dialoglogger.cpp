#include <cstdarg>
#include "dialoglogger.h"

void DialogLogger::push(const char* tag, const char* format, va_list ap) {
	char buf[MAX_LOG_SIZE];
	vsnprintf(buf, sizeof(buf), format, ap);
	std::string line = std::string(tag) + buf;
	lines_.push_back(line);
	if (echo_) fprintf(echo_, "%s\n", line.c_str());
}

#define LOGFN(name, tag)                        \
	void DialogLogger::name(const char* format, ...) { \
		va_list ap;                             \
		va_start(ap, format);                   \
		push(tag, format, ap);                  \
		va_end(ap);                             \
	}
LOGFN(info, "")
LOGFN(error, "[error] ")
LOGFN(warning, "[warning] ")
LOGFN(green, "[ok] ")

void DialogLogger::progress(int p) { progress_ = p; }
