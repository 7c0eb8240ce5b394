// Progress/log sink handed to the scene loader (reference: qt/include/dialoglogger.h:9-28,
// qt/dialoglogger.cpp:8-51 emit Qt signals to the load dialog). Headless version: same
// method names, lines are kept in memory and optionally echoed to a FILE*.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#define MAX_LOG_SIZE 1024

class DialogLogger {
public:
	explicit DialogLogger(FILE* echo = NULL) : echo_(echo), progress_(0) {}

	void info(const char* format, ...);
	void error(const char* format, ...);
	void warning(const char* format, ...);
	void green(const char* format, ...);
	void progress(int p);

	const std::vector<std::string>& lines() const { return lines_; }
	int lastProgress() const { return progress_; }

private:
	void push(const char* tag, const char* format, va_list ap);
	FILE* echo_;
	int progress_;
	std::vector<std::string> lines_;
};
