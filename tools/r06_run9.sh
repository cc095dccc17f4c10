mkdir -p gpurun_out/r06u
timeout 600 python tools/nms_probe.py > gpurun_out/r06u/nms_probe.txt 2>&1
timeout 900 python tools/copy_sources.py copyBuffer,fillBuffer,FillFunctor,copy_kernel,CUDAFunctor_add,CatArray,Memcpy,Memset,elementwise_kernel,reduce_kernel,gather,index,rocprim > gpurun_out/r06u/copy_sources.txt 2>&1
