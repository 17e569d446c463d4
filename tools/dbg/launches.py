import sys; sys.path.insert(0,'/root/repo')
import pathtrace_amd as pt
ctx=pt.Context(0); ctx.upload(pt.builtin_scene(2)); cam=pt.camera_new(width=1024,height=1024)
for i in range(2):
    ctx.render(cam, pt.default_params(spp=64, profile=1)); st=ctx.stats()
print("launches", st.bounce_launches, "total_ms", round(st.total_ms,3), "kernel_ms", round(st.bounce_kernel_ms,3), "primary_ms", round(st.primary_kernel_ms,3), "primary share", st.primary_vertices/st.vertices)
