/* Test-infrastructure shim: the reference's deps/GLEW/include/GL/glew.h is a
 * generated file (needs network); the navigation/movement TUs only need the GL
 * scalar typedefs (GLfloat ...) that pf_math.h pulls in. */
#ifndef PFREF_GLEW_SHIM_H
#define PFREF_GLEW_SHIM_H
#include <GL/gl.h>
#include <GL/glext.h>
#endif
