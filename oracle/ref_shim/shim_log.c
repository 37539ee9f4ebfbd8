#include <stdio.h>
int __android_log_write(int prio, const char *tag, const char *msg) {
  return fprintf(stderr, "[android-log %d] %s: %s\n", prio, tag ? tag : "", msg ? msg : "");
}
