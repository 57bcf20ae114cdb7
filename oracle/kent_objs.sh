#!/bin/sh
# Prints the member list of the reference's libcuskent.a: the "O = ..." object
# list of cuskent/makefile, read as data (the makefile itself is never run).
awk '/^O = /{f=1} f{line=$0; sub(/^O = /,"",line); gsub(/\\/,"",line); print line} f&&!/\\$/{exit}' "$1/cuskent/makefile"
