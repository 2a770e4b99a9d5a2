; The <2 x float> dataflow of rigid_step4_kernel<false> that SelectionDAG lowers wrongly in the whole kernel (tools/slp_repro/relower.py, stage 3:
; vector fadd/fsub/fmul number 85..92 of the kernel, value names as in its optimised IR), as a stand-alone kernel:
;   in  [rows][18] floats: a = %1150, b = %1151, c = %1157, p = %1389, q = %1429, r = %1469 (<2 x float> each), s0 = %1388, s1 = %1428, s2 = %1468, s3 = %1472, d = %1470
;   out [rows][4]  floats: %1482.x, %1482.y, %1490, %1495.x
target datalayout = "e-p:64:64-p1:64:64-p2:32:32-p3:32:32-p4:64:64-p5:32:32-p6:32:32-p7:160:256:256:32-p8:128:128:128:48-p9:192:256:256:32-i64:64-v16:16-v24:32-v32:32-v48:64-v96:128-v192:256-v256:256-v512:512-v1024:1024-v2048:2048-n32:64-S32-A5-G1-ni:7:8:9"
target triple = "amdgcn-amd-amdhsa"

declare i32 @llvm.amdgcn.workitem.id.x()
declare i32 @llvm.amdgcn.workgroup.id.x()

define amdgpu_kernel void @pk_repro(ptr addrspace(1) %in, ptr addrspace(1) %out) #0 {
  %tid = call i32 @llvm.amdgcn.workitem.id.x()
  %wg = call i32 @llvm.amdgcn.workgroup.id.x()
  %wg64 = shl i32 %wg, 6
  %row = add i32 %wg64, %tid
  %row64 = zext i32 %row to i64
  %ibase = mul i64 %row64, 18
  %pi = getelementptr float, ptr addrspace(1) %in, i64 %ibase
  %pa = getelementptr float, ptr addrspace(1) %pi, i64 0
  %pb = getelementptr float, ptr addrspace(1) %pi, i64 2
  %pc = getelementptr float, ptr addrspace(1) %pi, i64 4
  %pp = getelementptr float, ptr addrspace(1) %pi, i64 6
  %pq = getelementptr float, ptr addrspace(1) %pi, i64 8
  %pr = getelementptr float, ptr addrspace(1) %pi, i64 10
  %ps0 = getelementptr float, ptr addrspace(1) %pi, i64 12
  %ps1 = getelementptr float, ptr addrspace(1) %pi, i64 13
  %ps2 = getelementptr float, ptr addrspace(1) %pi, i64 14
  %ps3 = getelementptr float, ptr addrspace(1) %pi, i64 15
  %pd = getelementptr float, ptr addrspace(1) %pi, i64 16
  %v1150 = load <2 x float>, ptr addrspace(1) %pa, align 4
  %v1151 = load <2 x float>, ptr addrspace(1) %pb, align 4
  %v1157 = load <2 x float>, ptr addrspace(1) %pc, align 4
  %v1389 = load <2 x float>, ptr addrspace(1) %pp, align 4
  %v1429 = load <2 x float>, ptr addrspace(1) %pq, align 4
  %v1469 = load <2 x float>, ptr addrspace(1) %pr, align 4
  %v1388 = load float, ptr addrspace(1) %ps0, align 4
  %v1428 = load float, ptr addrspace(1) %ps1, align 4
  %v1468 = load float, ptr addrspace(1) %ps2, align 4
  %v1472 = load float, ptr addrspace(1) %ps3, align 4
  %v1470 = load <2 x float>, ptr addrspace(1) %pd, align 4
  %v1473 = insertelement <2 x float> %v1429, float %v1388, i64 0
  %v1474 = fmul contract <2 x float> %v1150, %v1473
  %v1475 = insertelement <2 x float> %v1389, float %v1428, i64 0
  %v1476 = fmul contract <2 x float> %v1157, %v1475
  %v1477 = fadd contract <2 x float> %v1474, %v1476
  %v1478 = shufflevector <2 x float> %v1151, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %v1479 = shufflevector <2 x float> %v1151, <2 x float> poison, <2 x i32> <i32 1, i32 1>
  %v1480 = insertelement <2 x float> %v1469, float %v1468, i64 0
  %v1481 = fmul contract <2 x float> %v1479, %v1480
  %v1482 = fadd contract <2 x float> %v1477, %v1481
  %v1483 = fmul contract <2 x float> %v1150, %v1389
  %v1484 = shufflevector <2 x float> %v1150, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %v1485 = fmul contract <2 x float> %v1484, %v1429
  %v1486 = fadd contract <2 x float> %v1483, %v1485
  %v1487 = shufflevector <2 x float> %v1151, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %v1488 = fmul contract <2 x float> %v1487, %v1469
  %v1489 = fadd contract <2 x float> %v1486, %v1488
  %v1490 = extractelement <2 x float> %v1489, i64 0
  %v1491 = insertelement <2 x float> %v1478, float %v1472, i64 1
  %v1492 = shufflevector <2 x float> %v1470, <2 x float> %v1150, <2 x i32> <i32 1, i32 3>
  %v1493 = fmul contract <2 x float> %v1491, %v1492
  %v1494 = shufflevector <2 x float> %v1493, <2 x float> poison, <2 x i32> <i32 1, i32 poison>
  %v1495 = fsub contract <2 x float> %v1493, %v1494
  %o0 = extractelement <2 x float> %v1482, i64 0
  %o1 = extractelement <2 x float> %v1482, i64 1
  %o3 = extractelement <2 x float> %v1495, i64 0
  %obase = mul i64 %row64, 4
  %po = getelementptr float, ptr addrspace(1) %out, i64 %obase
  %po1 = getelementptr float, ptr addrspace(1) %po, i64 1
  %po2 = getelementptr float, ptr addrspace(1) %po, i64 2
  %po3 = getelementptr float, ptr addrspace(1) %po, i64 3
  store float %o0, ptr addrspace(1) %po, align 4
  store float %o1, ptr addrspace(1) %po1, align 4
  store float %v1490, ptr addrspace(1) %po2, align 4
  store float %o3, ptr addrspace(1) %po3, align 4
  ret void
}

attributes #0 = { "amdgpu-flat-work-group-size"="1,64" "target-cpu"="gfx950" "uniform-work-group-size"="true" }
