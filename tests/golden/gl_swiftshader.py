"""A headless, conformant OpenGL ES 3.0 for the build container: Google SwiftShader (the CPU rasteriser Chromium ships),
found inside the `kaleido` wheel, driven through ctypes -- EGL 1.4 pbuffer surface, ES 3.0 context, no X server.

Only tests/golden/gen_gl_golden.py and tools/ reports use this module, in the BUILD CONTAINER.  Nothing under tests/ that runs
on the GPU box or in the CPU suite imports it: the renders travel as .npz fixtures.

What it offers is exactly what depth_map_tools.render (dmt:1422-1572) asks of Open3D's legacy Visualizer: draw one indexed
triangle list or one point list with per-vertex colours through `gl_Position = MVP * vec4(position, 1)`, GL_LESS depth test,
optional back-face culling, optional 4x multisampling, a clear colour; read the colour buffer back as RGBA8 and the depth
buffer as float (GL_NV_read_depth).
"""
import ctypes as C
import glob
import os

import numpy as np

EGL_NONE = 0x3038
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT = 0x3033, 0x0001
EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT = 0x3040, 0x0040
EGL_RED_SIZE, EGL_GREEN_SIZE, EGL_BLUE_SIZE, EGL_ALPHA_SIZE, EGL_DEPTH_SIZE = 0x3024, 0x3023, 0x3022, 0x3021, 0x3025
EGL_WIDTH, EGL_HEIGHT = 0x3057, 0x3056
EGL_CONTEXT_CLIENT_VERSION = 0x3098
EGL_OPENGL_ES_API = 0x30A0

GL_DEPTH_BUFFER_BIT, GL_COLOR_BUFFER_BIT = 0x0100, 0x4000
GL_POINTS, GL_TRIANGLES = 0x0000, 0x0004
GL_LESS = 0x0201
GL_CW, GL_CCW = 0x0900, 0x0901
GL_FRONT, GL_BACK = 0x0404, 0x0405
GL_CULL_FACE, GL_DEPTH_TEST, GL_DITHER, GL_BLEND = 0x0B44, 0x0B71, 0x0BD0, 0x0BE2
GL_PACK_ALIGNMENT = 0x0D05
GL_SUBPIXEL_BITS, GL_MAX_SAMPLES = 0x0D50, 0x8D57
GL_VENDOR, GL_RENDERER, GL_VERSION, GL_EXTENSIONS = 0x1F00, 0x1F01, 0x1F02, 0x1F03
GL_UNSIGNED_BYTE, GL_UNSIGNED_INT, GL_FLOAT = 0x1401, 0x1405, 0x1406
GL_DEPTH_COMPONENT, GL_RGBA = 0x1902, 0x1908
GL_NEAREST = 0x2600
GL_RGBA8, GL_DEPTH_COMPONENT24, GL_DEPTH_COMPONENT32F = 0x8058, 0x81A6, 0x8CAC
GL_ARRAY_BUFFER, GL_ELEMENT_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x8893, 0x88E4
GL_FRAGMENT_SHADER, GL_VERTEX_SHADER = 0x8B30, 0x8B31
GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B81, 0x8B82
GL_READ_FRAMEBUFFER, GL_DRAW_FRAMEBUFFER, GL_FRAMEBUFFER, GL_RENDERBUFFER = 0x8CA8, 0x8CA9, 0x8D40, 0x8D41
GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT, GL_FRAMEBUFFER_COMPLETE = 0x8CE0, 0x8D00, 0x8CD5

_VS = b"""#version 300 es
precision highp float;
uniform mat4 MVP;
in vec3 vertex_position;
in vec3 vertex_color;
out vec3 fragment_color;
void main() {
    gl_Position = MVP * vec4(vertex_position, 1.0);
    gl_PointSize = 1.0;
    fragment_color = vertex_color;
}
"""
_FS = b"""#version 300 es
precision highp float;
in vec3 fragment_color;
out vec4 FragColor;
void main() { FragColor = vec4(fragment_color, 1.0); }
"""


def swiftshader_dir():
    hits = glob.glob(os.path.join(os.path.dirname(os.__file__), "..", "..", "local", "lib", "python3*", "dist-packages", "kaleido",
                                  "executable", "bin", "swiftshader"))
    try:
        import kaleido
        hits.insert(0, os.path.join(os.path.dirname(kaleido.__file__), "executable", "bin", "swiftshader"))
    except ImportError:
        pass
    for h in hits:
        if os.path.exists(os.path.join(h, "libEGL.so")) and os.path.exists(os.path.join(h, "libGLESv2.so")):
            return os.path.abspath(h)
    return None


class GL:
    """One ES 3.0 context on a 1x1 pbuffer; all rendering goes to FBOs of the requested size."""

    def __init__(self):
        d = swiftshader_dir()
        if d is None:
            raise RuntimeError("SwiftShader (libEGL.so / libGLESv2.so of the kaleido wheel) not found")
        self.egl = egl = C.CDLL(os.path.join(d, "libEGL.so"))
        self.gl = gl = C.CDLL(os.path.join(d, "libGLESv2.so"))
        for f in ("eglGetDisplay", "eglCreatePbufferSurface", "eglCreateContext"):
            getattr(egl, f).restype = C.c_void_p
        egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        egl.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        egl.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        self.dpy = egl.eglGetDisplay(None)
        ma, mi = C.c_int(), C.c_int()
        if not egl.eglInitialize(self.dpy, C.byref(ma), C.byref(mi)):
            raise RuntimeError("eglInitialize failed")
        self.egl_version = f"{ma.value}.{mi.value}"
        egl.eglBindAPI(EGL_OPENGL_ES_API)
        attrs = (C.c_int * 15)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_RED_SIZE, 8,
                               EGL_GREEN_SIZE, 8, EGL_BLUE_SIZE, 8, EGL_DEPTH_SIZE, 24, EGL_NONE)
        cfg, n = C.c_void_p(), C.c_int()
        if not egl.eglChooseConfig(self.dpy, attrs, C.byref(cfg), 1, C.byref(n)) or n.value < 1:
            raise RuntimeError("eglChooseConfig found no ES3 pbuffer config")
        self.surf = egl.eglCreatePbufferSurface(self.dpy, cfg, (C.c_int * 5)(EGL_WIDTH, 1, EGL_HEIGHT, 1, EGL_NONE))
        self.ctx = egl.eglCreateContext(self.dpy, cfg, None, (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE))
        if not self.surf or not self.ctx or not egl.eglMakeCurrent(self.dpy, self.surf, self.surf, self.ctx):
            raise RuntimeError("EGL pbuffer / context creation failed")
        gl.glGetString.restype = C.c_char_p
        gl.glClearColor.argtypes = [C.c_float] * 4
        gl.glClearDepthf.argtypes = [C.c_float]
        gl.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
        gl.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
        gl.glDrawElements.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_void_p]
        gl.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        gl.glUniformMatrix4fv.argtypes = [C.c_int, C.c_int, C.c_ubyte, C.c_void_p]
        gl.glShaderSource.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_char_p), C.c_void_p]
        gl.glGetAttribLocation.argtypes = [C.c_uint, C.c_char_p]
        gl.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        self.version = gl.glGetString(GL_VERSION).decode()
        self.renderer = gl.glGetString(GL_RENDERER).decode()
        self.extensions = gl.glGetString(GL_EXTENSIONS).decode().split()
        self.subpixel_bits = self._geti(GL_SUBPIXEL_BITS)
        self.max_samples = self._geti(GL_MAX_SAMPLES)
        self.prog = self._program()
        self.loc_mvp = gl.glGetUniformLocation(self.prog, b"MVP")
        self.loc_pos = gl.glGetAttribLocation(self.prog, b"vertex_position")
        self.loc_col = gl.glGetAttribLocation(self.prog, b"vertex_color")

    def _geti(self, what):
        v = C.c_int()
        self.gl.glGetIntegerv(what, C.byref(v))
        return v.value

    def _check(self, where):
        e = self.gl.glGetError()
        if e:
            raise RuntimeError(f"GL error 0x{e:04x} at {where}")

    def _shader(self, kind, src):
        gl = self.gl
        s = gl.glCreateShader(kind)
        gl.glShaderSource(s, 1, C.byref(C.c_char_p(src)), None)
        gl.glCompileShader(s)
        ok = C.c_int()
        gl.glGetShaderiv(s, GL_COMPILE_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(4096)
            gl.glGetShaderInfoLog(s, 4096, None, log)
            raise RuntimeError("shader: " + log.value.decode())
        return s

    def _program(self):
        gl = self.gl
        p = gl.glCreateProgram()
        gl.glAttachShader(p, self._shader(GL_VERTEX_SHADER, _VS))
        gl.glAttachShader(p, self._shader(GL_FRAGMENT_SHADER, _FS))
        gl.glLinkProgram(p)
        ok = C.c_int()
        gl.glGetProgramiv(p, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            raise RuntimeError("program link failed")
        return p

    def meta(self):
        return (f"{self.renderer}; {self.version}; EGL {self.egl_version}; GL_SUBPIXEL_BITS {self.subpixel_bits}; "
                f"GL_MAX_SAMPLES {self.max_samples}")

    def _fbo(self, W, H, samples, depth_format):
        gl = self.gl
        fbo, rbs = C.c_uint(), (C.c_uint * 2)()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glGenRenderbuffers(2, rbs)
        for rb, fmt in ((rbs[0], GL_RGBA8), (rbs[1], depth_format)):
            gl.glBindRenderbuffer(GL_RENDERBUFFER, rb)
            if samples:
                gl.glRenderbufferStorageMultisample(GL_RENDERBUFFER, samples, fmt, W, H)
            else:
                gl.glRenderbufferStorage(GL_RENDERBUFFER, fmt, W, H)
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, rbs[0])
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rbs[1])
        if gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError("framebuffer incomplete")
        self._check("fbo")
        return fbo, rbs

    def render(self, W, H, positions, colors, MVP, triangles=None, bg=(0.0, 0.0, 0.0), cull=False, samples=0,
               depth_bits=24, read_depth=True, mirror_y=False):
        """positions f32[N,3], colors f32[N,3] in [0,1] (Open3D uploads both as float), MVP f32[4,4] (row-major here, i.e. as
        written on paper), triangles i32[M,3] or None for GL_POINTS.  Returns (rgba u8[H,W,4], depth f32[H,W] window z or None)
        with row 0 = TOP of the window (Open3D flips its read-back the same way).  samples > 0 renders to a multisampled FBO
        and resolves it with glBlitFramebuffer, which is what reading a multisampled window back does.

        mirror_y: SwiftShader applies its (Direct3D-style) top-left fill rule in FRAMEBUFFER MEMORY ORDER, and an FBO's memory
        starts with GL's bottom row -- so an upright render gives pixel centres that lie exactly on a horizontal edge to the
        triangle BELOW the edge in GL's y-up coordinates, i.e. (after the read-back flip) a "bottom-left" rule in the image.
        GL drivers that render to a window-system framebuffer (what Open3D reads back) store the TOP row first and apply the
        same hardware rule there: top-left in the image.  mirror_y negates clip-space y (and the front-face winding with it) and
        skips the read-back flip: the same picture, but memory order == image order, so that SwiftShader's rule acts in image
        space as it does on such a driver."""
        gl = self.gl
        positions = np.ascontiguousarray(positions, np.float32)
        colors = np.ascontiguousarray(colors, np.float32)
        depth_format = GL_DEPTH_COMPONENT24 if depth_bits == 24 else GL_DEPTH_COMPONENT32F
        fbo, rbs = self._fbo(W, H, samples, depth_format)
        gl.glViewport(0, 0, W, H)
        gl.glDisable(GL_DITHER); gl.glDisable(GL_BLEND)
        gl.glEnable(GL_DEPTH_TEST); gl.glDepthFunc(GL_LESS)
        if cull:
            gl.glEnable(GL_CULL_FACE); gl.glCullFace(GL_BACK); gl.glFrontFace(GL_CW if mirror_y else GL_CCW)
        else:
            gl.glDisable(GL_CULL_FACE)
        gl.glClearColor(float(bg[0]), float(bg[1]), float(bg[2]), 1.0)
        gl.glClearDepthf(1.0)
        gl.glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
        gl.glUseProgram(self.prog)
        MVP = np.array(MVP, np.float32)
        if mirror_y:
            MVP[1, :] = -MVP[1, :]
        m = np.ascontiguousarray(MVP.T)                                   # column-major for GL (ES forbids transpose=TRUE)
        gl.glUniformMatrix4fv(self.loc_mvp, 1, 0, m.ctypes.data)
        bufs = (C.c_uint * 3)()
        gl.glGenBuffers(3, bufs)
        for b, arr, loc in ((bufs[0], positions, self.loc_pos), (bufs[1], colors, self.loc_col)):
            gl.glBindBuffer(GL_ARRAY_BUFFER, b)
            gl.glBufferData(GL_ARRAY_BUFFER, arr.nbytes, arr.ctypes.data, GL_STATIC_DRAW)
            gl.glEnableVertexAttribArray(loc)
            gl.glVertexAttribPointer(loc, 3, GL_FLOAT, 0, 0, None)
        if triangles is None:
            gl.glDrawArrays(GL_POINTS, 0, positions.shape[0])
        else:
            idx = np.ascontiguousarray(triangles, np.uint32)
            gl.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, bufs[2])
            gl.glBufferData(GL_ELEMENT_ARRAY_BUFFER, idx.nbytes, idx.ctypes.data, GL_STATIC_DRAW)
            gl.glDrawElements(GL_TRIANGLES, idx.size, GL_UNSIGNED_INT, None)
        gl.glFinish()
        self._check("draw")
        read_fbo, read_rbs = fbo, None
        if samples:
            read_fbo, read_rbs = self._fbo(W, H, 0, depth_format)
            gl.glBindFramebuffer(GL_READ_FRAMEBUFFER, fbo)
            gl.glBindFramebuffer(GL_DRAW_FRAMEBUFFER, read_fbo)
            gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, GL_COLOR_BUFFER_BIT, GL_NEAREST)
            self._check("resolve")
            read_depth = False                                            # (a depth resolve is not defined the same way everywhere)
        gl.glBindFramebuffer(GL_FRAMEBUFFER, read_fbo)
        gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        rgba = np.zeros((H, W, 4), np.uint8)
        gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, rgba.ctypes.data)
        self._check("read colour")
        depth = None
        if read_depth and "GL_NV_read_depth" in self.extensions:
            depth = np.zeros((H, W), np.float32)
            gl.glReadPixels(0, 0, W, H, GL_DEPTH_COMPONENT, GL_FLOAT, depth.ctypes.data)
            if gl.glGetError():
                depth = None
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        for f, r in ((fbo, rbs), (read_fbo, read_rbs)):
            if r is not None:
                gl.glDeleteFramebuffers(1, C.byref(f))
                gl.glDeleteRenderbuffers(2, r)
        gl.glDeleteBuffers(3, bufs)
        if mirror_y:
            return rgba, depth
        return rgba[::-1].copy(), None if depth is None else depth[::-1].copy()
